#!/bin/bash
# round 5, call 31: config C with the sources of the last commit (tools/_old_tree, not committed) beside the working tree, same box:
# bench value twice each, then rocprofv3 kernel stats of one 7 500-alignment engine (2 passes) for both
out=$PWD/gpurun_out/r05_c31; mkdir -p $out
export TMPDIR=/tmp
run() {
  label=$1; dir=$2
  (cd $dir && timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg > $out/$label.json 2> $out/$label.err)
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
for rep in 1 2; do run new_$rep .; run old_$rep tools/_old_tree; done
for v in new old; do
  dir=.; [ $v = old ] && dir=tools/_old_tree
  (cd $dir && MPRG_BACKEND=runtime rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -- python3 tools/forest_profile.py 7500 2 > $out/run_$v.txt 2>&1)
  f=$(find $out/prof_$v -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_$v.csv; rm -rf $out/prof_$v
done
python - <<PY
import csv
def load(p):
    return {r["Name"].split("(")[0]: (int(r["Calls"]), float(r["TotalDurationNs"])/1e6) for r in csv.DictReader(open(p))}
a, b = load("$out/kernel_stats_old.csv"), load("$out/kernel_stats_new.csv")
rows = sorted(set(a) | set(b), key=lambda k: -max(a.get(k,(0,0))[1], b.get(k,(0,0))[1]))
for k in rows[:40]:
    print(f"{k[:60]:60s} old {a.get(k,(0,0))[0]:5d} {a.get(k,(0,0))[1]:8.2f}   new {b.get(k,(0,0))[0]:5d} {b.get(k,(0,0))[1]:8.2f}")
print("total old", sum(v[1] for v in a.values()), "new", sum(v[1] for v in b.values()))
PY

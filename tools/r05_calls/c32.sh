#!/bin/bash
# round 5, call 32: bench value, last commit's sources (tools/_old_tree, not committed) beside the working tree on one box, three times each
out=$PWD/gpurun_out/r05_c32; mkdir -p $out
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
for rep in 1 2 3; do run new_$rep .; run old_$rep tools/_old_tree; done

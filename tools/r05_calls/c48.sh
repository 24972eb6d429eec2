#!/bin/bash
# round 5, call 48: how much of a first pass's cost is the headroom of its predicted capacities?  3 750 / 7 500 / 15 000 alignments with
# MPRG_PLAN_HEAD 1.35 (the default) / 1.2 / 1.1 (misses counted), and the planned rate beside them
out=gpurun_out/r05_c48; mkdir -p $out
run() {
  label=$1; shift
  env "$@" timeout 500 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "misses", c.get("plan_misses_per_step"), "resumes", c.get("plan_resumes_per_step"), "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
for n in 3750 7500 15000; do
  for h in 1.35 1.2 1.1; do ARGS="--batch $n --first-pass" run f${n}_h$h MPRG_PLAN_HEAD=$h; done
  ARGS="--batch $n" run p$n X=1
done

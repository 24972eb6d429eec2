#!/bin/bash
# round 5, call 22: eight hardware queues by default: the 30 000-alignment step, shards, the command line
out=gpurun_out/r05_c22; mkdir -p $out
run() {
  label=$1; shift
  env "$@" timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "misses", c["plan_misses_per_step"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
ARGS="--batch 30000" run p30000_q8 X=1
ARGS="--batch 30000" run p30000_q4 GPU_MAX_HW_QUEUES=4
ARGS="--batch 7500 --first-pass" run f7500_q8 X=1
ARGS="--batch 7500 --first-pass" run f7500_q4 GPU_MAX_HW_QUEUES=4
ARGS="--batch 15000 --first-pass" run f15000_q8 X=1
ARGS="--batch 3750 --first-pass" run f3750_q8 X=1
ARGS="--batch 3750" run p3750_q8 X=1
timeout 900 python tools/cli_bench.py 30000 16 p a p:GPU_MAX_HW_QUEUES=4 a:GPU_MAX_HW_QUEUES=4 p a 2>&1 | grep -E "^-O"

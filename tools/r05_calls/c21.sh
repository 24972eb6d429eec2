#!/bin/bash
# round 5, call 21: first-pass rates of a 3 750-alignment shard by engines per worker x hardware queues x side streams
out=gpurun_out/r05_c21; mkdir -p $out
run() {
  label=$1; shift
  env "$@" timeout 300 python bench.py --batch 3750 --steps 8 --warmup 2 --first-pass --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "waits", c["host_waits_per_step"], "misses", c["plan_misses_per_step"], "streams", c["streams_per_worker"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
for q in 4 8; do for s in 2 3 4; do
  ARGS="--workers 1 --streams $s" run q${q}_s${s}_noside GPU_MAX_HW_QUEUES=$q MPRG_KM_SIDE_STREAMS=0
  ARGS="--workers 1 --streams $s" run q${q}_s${s}_side GPU_MAX_HW_QUEUES=$q MPRG_KM_SIDE_STREAMS=1
done; done
ARGS="--workers 1 --streams 4" run q16_s4_noside GPU_MAX_HW_QUEUES=16 MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 1 --streams 6" run q16_s6_noside GPU_MAX_HW_QUEUES=16 MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 2 --streams 2" run w2_s2_q8_noside GPU_MAX_HW_QUEUES=8 MPRG_KM_SIDE_STREAMS=0

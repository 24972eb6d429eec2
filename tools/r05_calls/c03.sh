#!/bin/bash
# round 5, call 3: a 3 750-alignment shard with more engines per worker and more hardware queues (GPU_MAX_HW_QUEUES)
out=gpurun_out/r05_c03; mkdir -p $out
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --batch 3750 --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "waits", c["host_waits_per_step"], "misses", c["plan_misses_per_step"], "workers", c["host_worker_processes_per_gpu"], "streams", c["streams_per_worker"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
for q in 4 8 16; do
  for s in 2 3 4 6; do
    ARGS="--workers 1 --streams $s" run q${q}_s${s}_side GPU_MAX_HW_QUEUES=$q MPRG_KM_SIDE_STREAMS=1
    ARGS="--workers 1 --streams $s" run q${q}_s${s}_noside GPU_MAX_HW_QUEUES=$q MPRG_KM_SIDE_STREAMS=0
  done
done
ARGS="--workers 2 --streams 2" run w2_s2_q8 GPU_MAX_HW_QUEUES=8 MPRG_KM_SIDE_STREAMS=1
ARGS="--workers 4 --streams 1" run w4_s1 MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 4 --streams 2" run w4_s2_q8 GPU_MAX_HW_QUEUES=8 MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 1 --streams 4 --first-pass" run q8_s4_first GPU_MAX_HW_QUEUES=8 MPRG_KM_SIDE_STREAMS=1
ARGS="--workers 1 --streams 4 --first-pass" run q16_s4_first GPU_MAX_HW_QUEUES=16 MPRG_KM_SIDE_STREAMS=1

#!/bin/bash
# round 5, call 25: more host shapes of the 30 000-alignment step with eight hardware queues
out=gpurun_out/r05_c25; mkdir -p $out
run() {
  label=$1; shift
  env "$@" timeout 500 python bench.py --batch 30000 --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "waits", c["host_waits_per_step"], "workers", c["host_worker_processes_per_gpu"], "streams", c["streams_per_worker"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
ARGS="--workers 4 --streams 1" run w4s1_rounds X=1
ARGS="--workers 1 --streams 4" run w1s4_rounds MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 1 --streams 8" run w1s8_fused MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 2 --streams 4" run w2s4_fused MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 2 --streams 2" run w2s2_rounds MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 4 --streams 2" run w4s2_fused MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 3 --streams 1" run w3s1_rounds X=1
ARGS="--workers 5 --streams 1" run w5s1_rounds X=1 MPRG_KLOOP=rounds

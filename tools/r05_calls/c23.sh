#!/bin/bash
# round 5, call 23: host shapes of the 7 500- and 15 000-alignment shards, first pass
out=gpurun_out/r05_c23; mkdir -p $out
run() {
  label=$1; shift
  env "$@" timeout 400 python bench.py --steps 8 --warmup 2 --first-pass --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "misses", c["plan_misses_per_step"], "workers", c["host_worker_processes_per_gpu"], "streams", c["streams_per_worker"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
for b in 7500 15000; do
ARGS="--batch $b --workers 4 --streams 1" run b${b}_w4s1 MPRG_KM_SIDE_STREAMS=0
ARGS="--batch $b --workers 4 --streams 1" run b${b}_w4s1_side MPRG_KM_SIDE_STREAMS=1
ARGS="--batch $b --workers 2 --streams 2" run b${b}_w2s2_side MPRG_KM_SIDE_STREAMS=1
ARGS="--batch $b --workers 2 --streams 2" run b${b}_w2s2 MPRG_KM_SIDE_STREAMS=0
ARGS="--batch $b --workers 1 --streams 4" run b${b}_w1s4 MPRG_KM_SIDE_STREAMS=0
ARGS="--batch $b --workers 1 --streams 2" run b${b}_w1s2_side MPRG_KM_SIDE_STREAMS=1
ARGS="--batch $b --workers 3 --streams 1" run b${b}_w3s1_side MPRG_KM_SIDE_STREAMS=1
done

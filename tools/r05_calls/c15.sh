#!/bin/bash
# round 5, call 15: host shapes for the 30 000-alignment step with this round's kernels (fused loop from plans against per-round launches)
out=gpurun_out/r05_c15; mkdir -p $out
run() {  # label, env..., args in ARGS
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
ARGS="--workers 4" run w4_rounds X=1
ARGS="--workers 4" run w4_fused MPRG_KLOOP=fused
ARGS="--workers 4 --streams 2" run w4_fused_s2 MPRG_KLOOP=fused MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 6" run w6_rounds X=1
ARGS="--workers 6" run w6_fused MPRG_KLOOP=fused
ARGS="--workers 8" run w8_rounds X=1
ARGS="--workers 8" run w8_fused MPRG_KLOOP=fused
ARGS="--workers 8 --streams 2" run w8_fused_s2 MPRG_KLOOP=fused MPRG_KM_SIDE_STREAMS=0
ARGS="--workers 4 --first-pass" run w4_fused_first MPRG_KLOOP=fused

#!/bin/bash
# round 5, call 5: fewer launches per level (one-launch scans with their capacity checks, one zero-fill): parity + small-shard rates
out=gpurun_out/r05_c05; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_speculative.py tests/test_gpu_parity.py tests/test_gpu_config_b.py -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -3 $out/pytest.txt
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "waits", c["host_waits_per_step"], "calls", c["launches_per_step"], "misses", c["plan_misses_per_step"], "workers", c["host_worker_processes_per_gpu"], "streams", c["streams_per_worker"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
for rep in 1 2; do
ARGS="--batch 3750" run p3750_$rep X=1
ARGS="--batch 3750 --first-pass" run f3750_$rep X=1
ARGS="--batch 3750 --streams 4" run p3750_s4q8_$rep GPU_MAX_HW_QUEUES=8 MPRG_KM_SIDE_STREAMS=0
ARGS="--batch 3750 --streams 4 --first-pass" run f3750_s4q8_$rep GPU_MAX_HW_QUEUES=8 MPRG_KM_SIDE_STREAMS=0
ARGS="--batch 7500" run p7500_$rep X=1
ARGS="--batch 7500 --first-pass" run f7500_$rep X=1
done
ARGS="--batch 30000" run p30000 X=1

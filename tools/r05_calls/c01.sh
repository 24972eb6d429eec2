#!/bin/bash
# round 5, call 1: GPU parity suite on the new sizing path + small-shard rates: planned vs first-pass
out=gpurun_out/r05_c01; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -3 $out/pytest.txt
for fp in "" "--first-pass"; do
  for b in 3750 7500; do
    timeout 600 python bench.py --batch $b --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $fp > $out/bench_${b}${fp}.json 2> $out/bench_${b}${fp}.err
    python - <<PY
import json
try:
    d=json.loads(open("$out/bench_${b}${fp}.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$b $fp", d["value"], d["ms_per_step"], "waits", c["host_waits_per_step"], "misses", c["plan_misses_per_step"], "resumes", c["plan_resumes_per_step"], "workers", c["host_worker_processes_per_gpu"], "streams", c["streams_per_worker"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$b $fp failed", e)
PY
  done
done

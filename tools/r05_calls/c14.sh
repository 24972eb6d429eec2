#!/bin/bash
# round 5, call 14: deep tests again; deep alignments' JSON with algorithmic bytes -> PMC summaries; first-pass rates after the headroom fix
out=gpurun_out/r05_c14; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_ddeep.py tests/test_gpu_parity.py -x -q -k "config_d_size or subsamples or many_workgroups" > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
for shape in "2000 4000" "5000 10000" "10000 20000"; do
  set -- $shape
  MPRG_DEEP_OUT=$out/deep_$1x$2.json timeout 1200 python tools/deep_profile.py $1 $2 7 --passes 3 > $out/deep_$1x$2.txt 2>&1
  grep -E '^\{' $out/deep_$1x$2.txt | tail -1 | cut -c1-200
done
for f in gpurun_out/r05_c13/rocprofv3_kernel_stats_2000x4000.csv gpurun_out/r05_c13/rocprofv3_kernel_stats_10000x20000.csv; do :; done
run() {  # label, args
  label=$1; shift
  MPRG_PLAN_TRACE=1 timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg "$@" > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "waits", c["host_waits_per_step"], "misses", c["plan_misses_per_step"], "resumes", c["plan_resumes_per_step"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
  grep -c "\[plan\]" $out/$label.err
}
run f15000 --batch 15000 --first-pass
run f7500 --batch 7500 --first-pass
run f3750 --batch 3750 --first-pass
run p3750 --batch 3750

#!/bin/bash
# round 6, call 24: the one-chunk views' three dedupe stages in k_ungap_dedupe's workgroup (MPRG_DD_ONE_CHUNK) against three launches
out=gpurun_out/r06_c24; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_ddeep.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_diagnostic_build_runs_the_fused_loops > $out/pytest_part.txt 2>&1; tail -3 $out/pytest_part.txt
for f in 1 0; do
  MPRG_DD_ONE_CHUNK=$f MPRG_PROFILE_ALL_LAUNCHES=1 timeout 600 python tools/forest_profile.py 7500 > $out/forest_7500_dd$f.txt 2>&1
  grep "per launch mprg_ungap_dedupe\|device time\|  mprg_ungap_dedupe\|  mprg_partition" $out/forest_7500_dd$f.txt
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for f in 1 0 1 0; do
  MPRG_DD_ONE_CHUNK=$f timeout 600 python bench.py $quick > $out/bench_dd${f}_$RANDOM.json 2> $out/bench_err.txt
  g=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$g').read().strip().splitlines()[-1]); print('30000 one-chunk $f:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

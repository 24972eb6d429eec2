#!/bin/bash
# round 6, call 50: the first centre.s cdf once per fit (a lane per sample, one division each) instead of D divisions walked by every restart.s lane — fc,
# against the committed build
out=gpurun_out/r06_c50; mkdir -p $out
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/make_prg_amd/_lib
MPRG_HIP_LIB=$L/libmprg_hip_fc.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_ddeep.py tests/test_kmeans_relocation.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_diagnostic_build_runs_the_fused_loops > $out/pytest_part.txt 2>&1; tail -3 $out/pytest_part.txt
for v in base fc base pair; do
  lib=$L/libmprg_hip_$v.so; [ $v = base ] && lib=$L/libmprg_hip.so
  MPRG_HIP_LIB=$lib timeout 600 python tools/forest_profile.py 7500 > $out/forest_7500_$v.txt 2>&1
  echo "== $v: $(grep '  mprg_kmeans_fit_lds' $out/forest_7500_$v.txt) | $(grep 'device time' $out/forest_7500_$v.txt)"
  MPRG_HIP_LIB=$lib timeout 600 python tools/forest_profile.py 3750 > $out/forest_3750_$v.txt 2>&1
  echo "   3750: $(grep '  mprg_cluster_loop.small' $out/forest_3750_$v.txt)"
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for v in base fc base pair; do
  lib=$L/libmprg_hip_$v.so; [ $v = base ] && lib=$L/libmprg_hip.so
  MPRG_HIP_LIB=$lib timeout 600 python bench.py $quick > $out/bench_${v}_$RANDOM.json 2> $out/bench_err.txt
  g=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$g').read().strip().splitlines()[-1]); print('30000 $v:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

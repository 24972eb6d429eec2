#!/bin/bash
# round 6, call 28: the centred rows as doubles in LDS beside the count bytes (KML_XD_BYTES: fits whose layout with them needs at most
# class 1 / 2 / 3's bytes take such a class), against the byte form alone (xd0)
out=gpurun_out/r06_c28; mkdir -p $out
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/make_prg_amd/_lib
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_ddeep.py tests/test_kmeans_relocation.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_diagnostic_build_runs_the_fused_loops > $out/pytest_part.txt 2>&1; tail -3 $out/pytest_part.txt
for v in xd0 xd1 xd2 xd3; do
  lib=$L/libmprg_hip_$v.so; [ $v = xd2 ] && lib=$L/libmprg_hip.so
  MPRG_HIP_LIB=$lib timeout 600 python tools/forest_profile.py 7500 > $out/forest_7500_$v.txt 2>&1
  echo "== $v"; grep "device time\|  mprg_kmeans_fit" $out/forest_7500_$v.txt
  MPRG_HIP_LIB=$lib timeout 600 python tools/forest_profile.py 3750 > $out/forest_3750_$v.txt 2>&1
  grep "device time\|  mprg_cluster_loop" $out/forest_3750_$v.txt
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for v in xd0 xd2 xd1 xd3 xd0 xd2 xd1 xd3; do
  lib=$L/libmprg_hip_$v.so; [ $v = xd2 ] && lib=$L/libmprg_hip.so
  MPRG_HIP_LIB=$lib timeout 600 python bench.py $quick > $out/bench_${v}_$RANDOM.json 2> $out/bench_err.txt
  g=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$g').read().strip().splitlines()[-1]); print('30000 $v:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

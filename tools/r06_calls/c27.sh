#!/bin/bash
# round 6, call 27: the one-workgroup partition (k_partition_fused: cells in LDS) for views of up to 16 / 24 KB of cells instead of 8 KB
out=gpurun_out/r06_c27; mkdir -p $out
export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/make_prg_amd/_lib
for v in pf16 pf24 base; do
  lib=$L/libmprg_hip_$v.so; [ $v = base ] && lib=$L/libmprg_hip.so
  MPRG_HIP_LIB=$lib MPRG_PROFILE_ALL_LAUNCHES=1 timeout 600 python tools/forest_profile.py 7500 > $out/forest_7500_$v.txt 2>&1
  echo "== $v"; grep "per launch mprg_partition\|device time\|  mprg_partition" $out/forest_7500_$v.txt
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for v in pf16 base pf24 pf16 base pf24; do
  lib=$L/libmprg_hip_$v.so; [ $v = base ] && lib=$L/libmprg_hip.so
  MPRG_HIP_LIB=$lib timeout 600 python bench.py $quick > $out/bench_${v}_$RANDOM.json 2> $out/bench_err.txt
  g=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$g').read().strip().splitlines()[-1]); print('30000 $v:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

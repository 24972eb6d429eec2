#!/bin/bash
# the deep (hierarchical) alignment: which launch form its big fits want.  Each setting: parity (the ddeep fixtures) + time
out=gpurun_out/r04_c24; mkdir -p $out
export TMPDIR=/tmp
run() {
  tag=$1; shift
  echo "=== $tag: $*"
  env "$@" timeout 900 python -m pytest tests/test_gpu_ddeep.py -x -q 2>&1 | tail -2
  env "$@" timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 1 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-200 | tee $out/deep_$tag.txt
}
run fused_1024 MPRG_KM_THREADS=1024
run rounds_256 MPRG_KLOOP=rounds
run rounds_1024 MPRG_KLOOP=rounds MPRG_KM_THREADS=1024
run rounds_split MPRG_KLOOP=rounds MPRG_KM_SPLIT_BELOW=1000000
timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 1 --check 200 2>&1 | tail -3

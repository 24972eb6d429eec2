#!/bin/bash
# deep alignment: split form with WIDE workgroups per restart
out=gpurun_out/r04_c26; mkdir -p $out
run() {
  tag=$1; shift
  echo "=== $tag: $*"
  env "$@" timeout 900 python -m pytest tests/test_gpu_ddeep.py -x -q 2>&1 | tail -2
  env "$@" timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 1 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-200 | tee $out/deep_$tag.txt
}
run split_256 MPRG_KLOOP=rounds MPRG_KM_SPLIT_BELOW=1000000 MPRG_KM_SPLIT_THREADS=256
run split_1024 MPRG_KLOOP=rounds MPRG_KM_SPLIT_BELOW=1000000 MPRG_KM_SPLIT_THREADS=1024

#!/bin/bash
# k-mer dictionaries of big problems by many workgroups: deep alignments + parity
out=gpurun_out/r04_c34; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ddeep.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for sz in "5000 10000" "10000 20000"; do
  timeout 1500 python tools/deep_profile.py $sz 7 --passes 1 --check 60 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-200 | tee $out/deep_${sz// /x}.txt
done

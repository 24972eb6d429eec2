#!/bin/bash
# deep alignments after the per-problem choice (tables below 160 MB, on demand above); full GPU tests; the full-size deep config D
out=gpurun_out/r04_c29; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $out/pytest_gpu.txt
MPRG_DEEP_OUT=$out/deep_2000x4000.json timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 2 2>&1 | grep -v amdgpu.ids | head -9 | cut -c1-200 | tee $out/deep_2000x4000.txt
MPRG_DEEP_OUT=$out/deep_5000x10000.json timeout 900 python tools/deep_profile.py 5000 10000 7 --passes 1 --check 60 2>&1 | grep -v amdgpu.ids | head -12 | cut -c1-200 | tee $out/deep_5000x10000.txt
MPRG_DEEP_OUT=$out/deep_10000x20000.json timeout 1500 python tools/deep_profile.py 10000 20000 7 --passes 1 --check 60 2>&1 | grep -v amdgpu.ids | head -14 | cut -c1-200 | tee $out/deep_10000x20000.txt

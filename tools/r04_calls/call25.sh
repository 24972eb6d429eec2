#!/bin/bash
# phase cycles of the KMeans fits of the deep alignment (diagnostic build; per-round launches)
out=gpurun_out/r04_c25; mkdir -p $out
MPRG_KLOOP=rounds timeout 900 python tools/phase_timing.py deep 2000 4000 2>&1 | grep -v amdgpu.ids | tee $out/phase_deep_2000x4000.txt

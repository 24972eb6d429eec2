#!/bin/bash
# phase cycles of the wide fits (diagnostic build) on two deep alignments
out=gpurun_out/r04_c30; mkdir -p $out
MPRG_KLOOP=rounds timeout 600 python tools/phase_timing.py deep 2000 4000 2>&1 | grep -v "amdgpu.ids\|k_partition" | tee $out/phase_wide_2000x4000.txt
MPRG_KLOOP=rounds timeout 900 python tools/phase_timing.py deep 5000 10000 2>&1 | grep -v "amdgpu.ids\|k_partition" | tee $out/phase_wide_5000x10000.txt

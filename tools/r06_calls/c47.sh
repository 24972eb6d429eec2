#!/bin/bash
# round 6, call 47: phase cycles per fit of the LDS form after the barrier / chain-phase merges (diagnostic build, per-round kernels, 4 096 alignments)
out=gpurun_out/r06_c47; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/phase_timing.py 4096 > $out/phases_4096.txt 2>&1; grep "%\|fits" $out/phases_4096.txt | head -16

#!/bin/bash
# hierarchical (deep) alignments per entry point: the parity-checked ddeep shape, then 5 000 x 10 000
out=gpurun_out/r04_c23; mkdir -p $out
export TMPDIR=/tmp
MPRG_DEEP_OUT=$out/deep_2000x4000.json timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 3 2>&1 | tee $out/deep_2000x4000.txt | cut -c1-220
MPRG_DEEP_OUT=$out/deep_5000x10000.json timeout 900 python tools/deep_profile.py 5000 10000 7 --passes 2 --check 40 2>&1 | tee $out/deep_5000x10000.txt | cut -c1-220

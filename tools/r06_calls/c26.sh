#!/bin/bash
# round 6, call 26: phases of the wide fits of one deep alignment (diagnostic build; 5 000 x 10 000 and 10 000 x 20 000)
out=gpurun_out/r06_c26; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python tools/phase_timing.py deep 5000 10000 > $out/phases_5000.txt 2>&1; grep "%\|fits" $out/phases_5000.txt | head -16
timeout 900 python tools/phase_timing.py deep 10000 20000 > $out/phases_10000.txt 2>&1; grep "%\|fits" $out/phases_10000.txt | head -16

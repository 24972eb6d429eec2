#!/bin/bash
# round 6, call 17: 30 000 fresh config-C alignments (seeds 2 000 000..), HIP path against the oracle, on the round's last kernels
out=gpurun_out/r06_c17; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python tools/parity_sweep.py 30000 2000000 > $out/sweep_config_c_30000.txt 2>&1; tail -2 $out/sweep_config_c_30000.txt | cut -c1-200

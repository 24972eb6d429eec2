#!/bin/bash
# round 6, call 33: parity sweeps of the round's last sources on fresh seeds (12 000 config-C alignments from seed 3 000 000, 6 x 1 500 nasty ones)
out=gpurun_out/r06_c33; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python tools/parity_sweep.py 12000 3000000 > $out/sweep_config_c_3000000.txt 2>&1; tail -2 $out/sweep_config_c_3000000.txt | cut -c1-200
timeout 1500 python tools/parity_sweep_nasty.py 1500 > $out/sweep_nasty_1500.txt 2>&1; tail -7 $out/sweep_nasty_1500.txt | cut -c1-200

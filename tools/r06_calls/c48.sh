#!/bin/bash
# round 6, call 48: parity sweeps of the round's LAST sources on fresh seeds (30 000 config-C alignments from seed 4 000 000, 6 x 1 500 nasty ones)
out=gpurun_out/r06_c48; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python tools/parity_sweep.py 30000 4000000 > $out/sweep_config_c_4000000.txt 2>&1; tail -2 $out/sweep_config_c_4000000.txt | cut -c1-200
timeout 1500 python tools/parity_sweep_nasty.py 1500 > $out/sweep_nasty_1500.txt 2>&1; tail -7 $out/sweep_nasty_1500.txt | cut -c1-200

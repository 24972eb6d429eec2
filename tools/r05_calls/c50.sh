#!/bin/bash
# round 5, call 50: larger sweeps against the oracle on the round's last sources — 30 000 fresh config-C seeds, 6 x 1 500 nasty alignments,
# hierarchical alignments with every level counted as big (wide fits, tiled tables, predict launches, wide majority / hamming / dedupe)
out=gpurun_out/r05_c50; mkdir -p $out
timeout 1700 python tools/parity_sweep.py 30000 1000000 > $out/sweep_config_c_30000.txt 2>&1; tail -2 $out/sweep_config_c_30000.txt | cut -c1-200
timeout 1700 python tools/parity_sweep_nasty.py 1500 > $out/sweep_nasty_1500.txt 2>&1; tail -6 $out/sweep_nasty_1500.txt | cut -c1-200
timeout 1700 python tools/deep_sweep.py 8 400 900 4096 > $out/sweep_deep.txt 2>&1; tail -3 $out/sweep_deep.txt | cut -c1-200
timeout 1700 python tools/deep_sweep.py 3 1300 1500 65536 > $out/sweep_deep_tall.txt 2>&1; tail -2 $out/sweep_deep_tall.txt | cut -c1-200

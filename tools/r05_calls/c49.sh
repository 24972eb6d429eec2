#!/bin/bash
# round 5, call 49: parity sweeps outside the suite on the round's last sources: fresh config-C seeds and the nasty generator, HIP path vs oracle
out=gpurun_out/r05_c49; mkdir -p $out
timeout 1500 python tools/parity_sweep.py 3000 700000 > $out/sweep_config_c.txt 2>&1; tail -3 $out/sweep_config_c.txt | cut -c1-200
timeout 1500 python tools/parity_sweep_nasty.py 400 > $out/sweep_nasty.txt 2>&1; tail -4 $out/sweep_nasty.txt | cut -c1-200

#!/bin/bash
# round 5, call 39: the diagnostic build on config C (fused loops) and on the deep alignment with the per-round loop
out=gpurun_out/r05_c39; mkdir -p $out
timeout 600 python tools/phase_timing.py 2048 > $out/phase_c.txt 2>&1; echo config C; tail -6 $out/phase_c.txt | cut -c1-160
MPRG_KLOOP=rounds timeout 900 python tools/phase_timing.py deep 2000 4000 > $out/phase_deep_rounds.txt 2>&1; echo deep rounds; grep -v "k_partition" $out/phase_deep_rounds.txt | tail -18 | cut -c1-200
MPRG_KLOOP=rounds timeout 900 python tools/phase_timing.py deep 10000 20000 > $out/phase_deep10k_rounds.txt 2>&1; echo deep 10k rounds; grep -v "k_partition" $out/phase_deep10k_rounds.txt | tail -18 | cut -c1-200

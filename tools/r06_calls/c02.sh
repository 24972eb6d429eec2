#!/bin/bash
# round 6, call 2: where a fit's cycles go — diagnostic build (-DKM_PHASE_TIMING), per-round loop, forms of round 5 (MPRG_KM_MODE=2) and the LDS
# form (6); then the diagnostic build's FUSED loops (round 5's faulted on the device: ADVICE r05)
out=gpurun_out/r06_c02; mkdir -p $out
export TMPDIR=/tmp
for mode in 2 6; do
  MPRG_KM_MODE=$mode timeout 600 python tools/phase_timing.py 4096 > $out/phase_rounds_mode$mode.txt 2>&1
  grep -v "k_partition" $out/phase_rounds_mode$mode.txt | cut -c1-160
done
MPRG_KLOOP=fused MPRG_KM_MODE=6 timeout 600 python tools/phase_timing.py 2048 > $out/phase_fused_mode6.txt 2>&1; echo "fused lds rc=$?"; grep -v "k_partition" $out/phase_fused_mode6.txt | tail -18 | cut -c1-160
MPRG_KLOOP=fused MPRG_KM_MODE=2 timeout 600 python tools/phase_timing.py 2048 > $out/phase_fused_mode2.txt 2>&1; echo "fused small rc=$?"; tail -5 $out/phase_fused_mode2.txt | cut -c1-200

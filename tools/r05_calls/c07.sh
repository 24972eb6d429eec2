#!/bin/bash
# round 5, call 7: per-level launch times of the byte kernels, wave forms on / off (7 500 alignments, one process, exclusive)
out=gpurun_out/r05_c07; mkdir -p $out
for wv in 1 0; do
  MPRG_WAVE_VIEWS=$wv MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_wv$wv.txt 2>&1
  grep -E "per launch|device time|mprg_" $out/profile_wv$wv.txt | head -40
done

#!/bin/bash
# round 5, call 24: per-launch times of the wide fits of the deep alignments (which level holds the time)
out=gpurun_out/r05_c24; mkdir -p $out
for shape in "2000 4000" "10000 20000"; do
  set -- $shape
  MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_PLAN_TRACE=1 timeout 900 python tools/deep_profile.py $1 $2 7 --passes 1 > $out/deep_$1x$2.txt 2>&1
  grep -E "per launch|^\{" $out/deep_$1x$2.txt | cut -c1-400
done

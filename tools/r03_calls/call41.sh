#!/bin/bash
mkdir -p gpurun_out/r03_c41
for n in 3750 7500 30000; do for s in 0 1; do
  MPRG_BACKEND=runtime MPRG_KM_SIDE_STREAMS=$s python tools/forest_profile.py $n 3 > gpurun_out/r03_c41/prof_${n}_side$s.txt 2>&1
  echo "== $n side $s"; tail -3 gpurun_out/r03_c41/prof_${n}_side$s.txt
done; done

#!/bin/bash
mkdir -p gpurun_out/r03_c35
for b in torch runtime; do
  MPRG_BACKEND=$b python tools/forest_profile.py 30000 3 > gpurun_out/r03_c35/prof_$b.txt 2>&1
  echo "== $b"; grep "^step\|device time" gpurun_out/r03_c35/prof_$b.txt; tail -5 gpurun_out/r03_c35/prof_$b.txt
  MPRG_BACKEND=$b python tools/forest_profile.py 3750 3 > gpurun_out/r03_c35/prof3750_$b.txt 2>&1
  tail -3 gpurun_out/r03_c35/prof3750_$b.txt
done

#!/bin/bash
# KMeans kernels compiled for 4 / 3 / 2 waves per SIMD (register budget 128 / 170 / 256: spills vs occupancy), small and general form apart
mkdir -p gpurun_out/r03_c32
for tag in base s3g4 s4g3 s3g3 s2g2; do
  lib=make_prg_amd/_lib/libmprg_hip_$tag.so; [ $tag = base ] && lib=make_prg_amd/_lib/libmprg_hip.so
  MPRG_HIP_LIB=$PWD/$lib python tools/forest_profile.py 30000 2 > gpurun_out/r03_c32/prof_$tag.txt 2>&1
  echo "== $tag"; grep "step 1\|kmeans_fit\|device time" gpurun_out/r03_c32/prof_$tag.txt
done

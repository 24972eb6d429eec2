#!/bin/bash
# round 5, call 34: a test-only build whose small KMeans form keeps a 4 KB pool (fits of at most ten sequences qualify; 10.7 KB of LDS per
# workgroup): two wavefronts per fit (8 fits per CU, register-bound) beside one wavefront per fit (15 fits per CU) on the same fits
out=gpurun_out/r05_c34; mkdir -p $out
L=$PWD/make_prg_amd/_lib/libmprg_hip_pool4k.so
for t in 128 64; do
  MPRG_HIP_LIB=$L MPRG_KMS_THREADS=$t MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_kms$t.txt 2>&1
  echo threads $t; grep -E "per launch mprg_kmeans_fit_small|per launch mprg_kmeans_fit:|device time|mprg_kmeans_fit_small |mprg_kmeans_fit " $out/profile_kms$t.txt | cut -c1-230
done

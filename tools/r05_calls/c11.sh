#!/bin/bash
# round 5, call 11: threshold of the all-rounds-at-once launch; 10 000 x 20 000
out=gpurun_out/r05_c11; mkdir -p $out
for sp in 20 60 100000; do
  MPRG_KM_SPEC_PROBLEMS=$sp MPRG_DEEP_OUT=$out/deep_2000x4000_spec$sp.json timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 2 > $out/deep_2000x4000_spec$sp.txt 2>&1
  echo "2000x4000 spec $sp"; grep -E '^\{' $out/deep_2000x4000_spec$sp.txt | tail -1 | cut -c1-200; grep -E 'mprg_kmeans_fit_wide|mprg_kmeans_prepare_big|mprg_cluster_further|mprg_kmeans_fit ' $out/deep_2000x4000_spec$sp.txt
  MPRG_KM_SPEC_PROBLEMS=$sp MPRG_DEEP_OUT=$out/deep_5000x10000_spec$sp.json timeout 900 python tools/deep_profile.py 5000 10000 7 --passes 1 > $out/deep_5000x10000_spec$sp.txt 2>&1
  echo "5000x10000 spec $sp"; grep -E '^\{' $out/deep_5000x10000_spec$sp.txt | tail -1 | cut -c1-200; grep -E 'mprg_kmeans_fit_wide|mprg_kmeans_prepare_big' $out/deep_5000x10000_spec$sp.txt
done
for sp in 20 100000; do
  MPRG_KM_SPEC_PROBLEMS=$sp MPRG_DEEP_OUT=$out/deep_10000x20000_spec$sp.json timeout 1200 python tools/deep_profile.py 10000 20000 7 --passes 1 > $out/deep_10000x20000_spec$sp.txt 2>&1
  echo "10000x20000 spec $sp"; grep -E '^\{' $out/deep_10000x20000_spec$sp.txt | tail -1 | cut -c1-200; head -30 $out/deep_10000x20000_spec$sp.txt | grep -E "mprg_" | head -8
done

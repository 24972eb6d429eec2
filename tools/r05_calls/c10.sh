#!/bin/bash
# round 5, call 10: every round of a big level's KMeans at once (KM_SPEC_PROBLEMS): deep alignments, parity + wall
out=gpurun_out/r05_c10; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_ddeep.py -x -q > $out/pytest_ddeep.txt 2>&1; tail -3 $out/pytest_ddeep.txt
for sp in 0 5 20; do
  MPRG_KM_SPEC_PROBLEMS=$sp MPRG_DEEP_OUT=$out/deep_2000x4000_spec$sp.json timeout 600 python tools/deep_profile.py 2000 4000 7 --passes 2 > $out/deep_2000x4000_spec$sp.txt 2>&1
  grep -E '^\{|mprg_kmeans_fit_wide|mprg_kmeans_prepare_big' $out/deep_2000x4000_spec$sp.txt | cut -c1-260
done
for sp in 0 5 20; do
  MPRG_KM_SPEC_PROBLEMS=$sp MPRG_DEEP_OUT=$out/deep_5000x10000_spec$sp.json timeout 900 python tools/deep_profile.py 5000 10000 7 --passes 1 > $out/deep_5000x10000_spec$sp.txt 2>&1
  grep -E '^\{|mprg_kmeans_fit_wide|mprg_kmeans_prepare_big' $out/deep_5000x10000_spec$sp.txt | cut -c1-260
done

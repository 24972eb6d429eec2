#!/bin/bash
# round 6, call 12: parity sweeps of the round's kernels on fresh seeds (HIP path against the oracle) and the deep alignments' timings
out=gpurun_out/r06_c12; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python tools/parity_sweep.py 12000 1200000 > $out/sweep_config_c.txt 2>&1; tail -2 $out/sweep_config_c.txt | cut -c1-200
timeout 1500 python tools/parity_sweep_nasty.py 1500 > $out/sweep_nasty_1500.txt 2>&1; tail -7 $out/sweep_nasty_1500.txt | cut -c1-200
for shape in "2000 4000" "5000 10000" "10000 20000"; do
  set -- $shape
  MPRG_DEEP_OUT=$out/deep_$1x$2.json timeout 1200 python tools/deep_profile.py $1 $2 7 --passes 3 > $out/deep_$1x$2.txt 2>&1
  grep -E '^\{' $out/deep_$1x$2.txt | tail -1 | cut -c1-400
  grep -E "mprg_kmeans_fit_wide|mprg_kmeans_prepare_big|mprg_kmeans_fit_lds|mprg_kmeans_fit " $out/deep_$1x$2.txt | head -5
done

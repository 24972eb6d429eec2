#!/bin/bash
# round 6, call 1: the LDS form of the small KMeans fits (k_kmeans_fit_lds) on hardware — parity of the KMeans entry points, then per-entry-point
# device time of a 7 500-alignment forest (per-round clustering loop) and the bench value, with the forms of round 5 (MPRG_KM_MODE=2) beside it
out=gpurun_out/r06_c01; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kmeans or relocation" > $out/pytest_kmeans.txt 2>&1; tail -3 $out/pytest_kmeans.txt
for mode in 2 6; do
  MPRG_KM_MODE=$mode timeout 600 python tools/forest_profile.py 7500 3 > $out/profile_mode$mode.txt 2>&1
  grep -E "device time|mprg_kmeans|per launch mprg_kmeans_fit" $out/profile_mode$mode.txt | cut -c1-400
done
for mode in 2 6; do
  MPRG_KM_MODE=$mode timeout 600 python tools/forest_profile.py 3750 4 > $out/profile3750_mode$mode.txt 2>&1
  grep -E "^step|device time|mprg_cluster_loop|mprg_kmeans" $out/profile3750_mode$mode.txt | cut -c1-300
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for mode in 2 6 2 6; do
  MPRG_KM_MODE=$mode timeout 600 python bench.py $quick > $out/bench_mode${mode}_$RANDOM.json 2> $out/bench_err.txt
  tail -1 $out/bench_err.txt | cut -c1-200
done
for mode in 2 6; do
  MPRG_KM_MODE=$mode timeout 600 python bench.py $quick --batch 3750 --workers 1 --first-pass > $out/bench3750_mode${mode}.json 2> $out/bench_err.txt
  tail -1 $out/bench_err.txt | cut -c1-200
done
for f in $out/bench_mode*.json $out/bench3750*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('verified'), d['roofline'])"; done

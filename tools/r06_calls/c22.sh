#!/bin/bash
# round 6, call 22: a round's cluster_further over the round's launch lists (mprg_cluster_further_listed) against over the level's problems
out=gpurun_out/r06_c22; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py tests/test_gpu_ddeep.py -m gpu -x -q > $out/pytest_part.txt 2>&1; tail -3 $out/pytest_part.txt
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for cf in 1 0 1 0; do
  MPRG_KM_CF_LISTED=$cf timeout 600 python bench.py $quick > $out/bench_cf${cf}_$RANDOM.json 2> $out/bench_err.txt
  f=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('30000 listed $cf:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done
for cf in 1 0; do
  MPRG_KM_CF_LISTED=$cf timeout 600 python tools/forest_profile.py 7500 > $out/forest_7500_cf$cf.txt 2>&1; grep -i "device time\|kmeans_fit\|further\|advance" $out/forest_7500_cf$cf.txt | grep -v "per launch" | head -8
done

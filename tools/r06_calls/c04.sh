#!/bin/bash
# round 6, call 4: the LDS form after the M-step by four features / reciprocal table / distance_next_center table / per-sample pair lists:
# instruction counters (fused loop, 4 096 alignments), per-entry-point times at 7 500 (rounds) and 3 750 (fused), bench value
out=gpurun_out/r06_c04; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kmeans or relocation" > $out/pytest_kmeans.txt 2>&1; tail -1 $out/pytest_kmeans.txt
MODES=6 bash tools/r06_calls/c03.sh > $out/c03.txt 2>&1; cp gpurun_out/r06_c03/counters.txt $out/counters.txt
for n in 7500 3750; do
  MPRG_KM_MODE=6 timeout 600 python tools/forest_profile.py $n 3 > $out/profile${n}_mode6.txt 2>&1
  grep -E "device time|mprg_kmeans|mprg_cluster_loop" $out/profile${n}_mode6.txt | grep -v "per launch" | cut -c1-200
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for mode in 6 2 6; do
  MPRG_KM_MODE=$mode timeout 600 python bench.py $quick > $out/bench_mode${mode}_$RANDOM.json 2> $out/bench_err.txt
done
MPRG_KM_MODE=6 timeout 600 python bench.py $quick --batch 3750 --workers 1 --first-pass > $out/bench3750_mode6.json 2> $out/bench_err.txt
for f in $out/bench_mode*.json $out/bench3750*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], d['roofline']['kernel'], d['roofline']['frac'], [(k['entry_point'], k['ms']) for k in d['roofline']['kernels'][:3]])"; done

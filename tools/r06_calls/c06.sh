#!/bin/bash
# round 6, call 6: after the pair lists by wavefront slots (1 024 entries, several passes beyond), sixteen operands per trip along centre rows and the
# fifth LDS class: parity of the KMeans entry points, phase cycles, per-entry-point times (7 500 rounds / 3 750 fused), instruction counters, bench
out=gpurun_out/r06_c06; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kmeans or relocation" > $out/pytest_kmeans.txt 2>&1; tail -1 $out/pytest_kmeans.txt
MPRG_KM_MODE=6 timeout 600 python tools/phase_timing.py 4096 > $out/phase_rounds_mode6.txt 2>&1
grep -v "k_partition" $out/phase_rounds_mode6.txt | cut -c1-160
for n in 7500 3750; do
  timeout 600 python tools/forest_profile.py $n 3 > $out/profile${n}.txt 2>&1
  grep -E "device time|mprg_kmeans|mprg_cluster_loop" $out/profile${n}.txt | grep -v "per launch" | cut -c1-200
done
rm -f gpurun_out/r06_c03/counters.txt; MODES=6 bash tools/r06_calls/c03.sh > $out/c03.txt 2>&1; cp gpurun_out/r06_c03/counters.txt $out/counters.txt
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for i in 1 2; do
  timeout 600 python bench.py $quick > $out/bench_$i.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench_$i.json').read().strip().splitlines()[-1]); print('30000:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], d['roofline']['frac'], [(k['entry_point'], k['ms']) for k in d['roofline']['kernels'][:3]])"
done
for shape in "1 2" "2 1"; do
  set -- $shape
  timeout 600 python bench.py $quick --batch 3750 --workers $1 --streams $2 --first-pass > $out/bench3750_w$1_s$2.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench3750_w$1_s$2.json').read().strip().splitlines()[-1]); print('3750 first pass workers $1 engines $2:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

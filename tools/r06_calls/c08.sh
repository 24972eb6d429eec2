#!/bin/bash
# round 6, call 8: counts staged once per problem in the fused loop, the seeding's searches in parallel; fused against per-round loops at 7 500 per worker
out=gpurun_out/r06_c08; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/pytest_parity.txt 2>&1; tail -1 $out/pytest_parity.txt
MPRG_KM_MODE=6 timeout 600 python tools/phase_timing.py 4096 > $out/phase_rounds_mode6.txt 2>&1
grep -v "k_partition" $out/phase_rounds_mode6.txt | cut -c1-160
MPRG_KLOOP=fused MPRG_KM_MODE=6 timeout 600 python tools/phase_timing.py 2048 > $out/phase_fused_mode6.txt 2>&1
grep -v "k_partition" $out/phase_fused_mode6.txt | cut -c1-160
for n in 7500 3750; do
  timeout 600 python tools/forest_profile.py $n 3 > $out/profile${n}.txt 2>&1
  grep -E "device time|mprg_kmeans|mprg_cluster_loop" $out/profile${n}.txt | grep -v "per launch" | cut -c1-200
done
MPRG_KLOOP=fused timeout 600 python tools/forest_profile.py 7500 3 > $out/profile7500_fused.txt 2>&1
grep -E "device time|mprg_kmeans|mprg_cluster_loop" $out/profile7500_fused.txt | grep -v "per launch" | cut -c1-200
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for loop in auto fused auto fused; do
  MPRG_KLOOP=$loop timeout 600 python bench.py $quick > $out/bench_$loop_$RANDOM.json 2> $out/bench_err.txt
  f=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('30000 loop $loop:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], d['roofline']['frac'], [(k['entry_point'], k['ms']) for k in d['roofline']['kernels'][:3]])"
done
for shape in "1 2" "2 1"; do
  set -- $shape
  timeout 600 python bench.py $quick --batch 3750 --workers $1 --streams $2 --first-pass > $out/bench3750_w$1_s$2.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench3750_w$1_s$2.json').read().strip().splitlines()[-1]); print('3750 first pass workers $1 engines $2:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

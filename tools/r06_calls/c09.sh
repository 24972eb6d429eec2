#!/bin/bash
# round 6, call 9: LDS classes re-cut for the body's real static LDS (8 / 6 / 4 / 3 / 2 workgroups per CU); threads of the middle classes
out=gpurun_out/r06_c09; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kmeans or relocation" > $out/pytest_kmeans.txt 2>&1; tail -1 $out/pytest_kmeans.txt
for n in 7500 3750; do
  timeout 600 python tools/forest_profile.py $n 3 > $out/profile${n}.txt 2>&1
  grep -E "device time|mprg_kmeans|mprg_cluster_loop" $out/profile${n}.txt | grep -v "per launch" | cut -c1-200
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for cfg in "128 256 256" "256 256 256" "128 128 256" "128 256 256"; do
  set -- $cfg
  MPRG_KML_THREADS1=$1 MPRG_KML_THREADS2=$2 MPRG_KML_THREADS3=$3 timeout 600 python bench.py $quick > $out/bench_t_$1_$2_$3_$RANDOM.json 2> $out/bench_err.txt
  f=$(ls -t $out/bench_t_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('30000 threads c1 $1 c2 $2 c3 $3:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], d['roofline']['frac'], [(k['entry_point'], k['ms']) for k in d['roofline']['kernels'][:3]])"
done
for shape in "1 2" "2 1"; do
  set -- $shape
  timeout 600 python bench.py $quick --batch 3750 --workers $1 --streams $2 --first-pass > $out/bench3750_w$1_s$2.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench3750_w$1_s$2.json').read().strip().splitlines()[-1]); print('3750 first pass workers $1 engines $2:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

#!/bin/bash
# round 6, call 5: phase cycles of the LDS form after the instruction-count work; the shard of an 8-GPU rank (3 750 alignments, first pass) with the general
# form beside the LDS classes on a side stream, by host shape; the bench value
out=gpurun_out/r06_c05; mkdir -p $out
export TMPDIR=/tmp
MPRG_KM_MODE=6 timeout 600 python tools/phase_timing.py 4096 > $out/phase_rounds_mode6.txt 2>&1
grep -v "k_partition" $out/phase_rounds_mode6.txt | cut -c1-160
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for shape in "1 2" "1 4" "2 1" "4 1" "2 2"; do
  set -- $shape
  timeout 600 python bench.py $quick --batch 3750 --workers $1 --streams $2 --first-pass > $out/bench3750_w$1_s$2.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench3750_w$1_s$2.json').read().strip().splitlines()[-1]); print('3750 first pass workers $1 engines $2:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], d['config'].get('km_side_streams'))"
done
for side in 0 1; do
  MPRG_KM_SIDE_STREAMS=$side timeout 600 python bench.py $quick > $out/bench_side$side.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench_side$side.json').read().strip().splitlines()[-1]); print('30000 side $side:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], [(k['entry_point'], k['ms']) for k in d['roofline']['kernels'][:3]])"
done
# the centred rows from the workspace instead of the counts as bytes in LDS (MPRG_KML_FLAGS=1): fewer instructions per feature, more loads
for fl in 1; do
  MPRG_KML_FLAGS=$fl timeout 600 python bench.py $quick > $out/bench_flags$fl.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench_flags$fl.json').read().strip().splitlines()[-1]); print('30000 flags $fl:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], [(k['entry_point'], k['ms']) for k in d['roofline']['kernels'][:3]])"
  MPRG_KML_FLAGS=$fl timeout 600 python bench.py $quick --batch 3750 --workers 1 --first-pass > $out/bench3750_flags$fl.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench3750_flags$fl.json').read().strip().splitlines()[-1]); print('3750 flags $fl:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done
# threads per fit of the heavy classes
for t in "128 128" "256 128" "128 256"; do
  set -- $t
  MPRG_KML_THREADS2=$1 MPRG_KML_THREADS3=$2 timeout 600 python bench.py $quick > $out/bench_thr_$1_$2.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench_thr_$1_$2.json').read().strip().splitlines()[-1]); print('30000 threads c2 $1 c3 $2:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], [(k['entry_point'], k['ms']) for k in d['roofline']['kernels'][:3]])"
done

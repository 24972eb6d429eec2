#!/bin/bash
# round 6, call 13: host worker processes per GPU for the full job (30 000 alignments), with the LDS form
out=gpurun_out/r06_c13; mkdir -p $out
export TMPDIR=/tmp
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for w in 4 5 6 8 4 6; do
  timeout 600 python bench.py $quick --workers $w > $out/bench_w${w}_$RANDOM.json 2> $out/bench_err.txt
  f=$(ls -t $out/bench_w*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('30000 workers $w:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], d['config']['host_worker_processes_per_gpu'])"
done
for w in 6; do
  MPRG_KLOOP=rounds timeout 600 python bench.py $quick --workers $w > $out/bench_rounds_w${w}.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench_rounds_w${w}.json').read().strip().splitlines()[-1]); print('30000 workers $w, per-round loops:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

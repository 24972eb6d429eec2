#!/bin/bash
# round 6, call 20: the fused clustering loop for big batches, by number of worker processes (the per-round loop is the default from 6 000 alignments per engine on)
out=gpurun_out/r06_c20; mkdir -p $out
export TMPDIR=/tmp
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for cfg in "auto 4" "fused 4" "fused 6" "fused 8" "fused 5" "auto 4"; do
  set -- $cfg
  MPRG_KLOOP=$1 timeout 600 python bench.py $quick --workers $2 > $out/bench_$1_w$2_$RANDOM.json 2> $out/bench_err.txt
  f=$(ls -t $out/bench_*.json | head -1)
  python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('30000 loop $1 workers $2:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

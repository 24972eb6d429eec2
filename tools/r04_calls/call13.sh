#!/bin/bash
# do the engines' streams of one process overlap on the device?  kernel trace of the in-process bench, 1 / 2 / 4 streams
out=gpurun_out/r04_c13; mkdir -p $out
export TMPDIR=/tmp
for S in 1 2 4; do
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_s$S -- python3 bench.py --workers 0 --streams $S --batch 3750 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg > $out/bench_s$S.json 2> $out/err_s$S.txt
  f=$(find $out/trace_s$S -name "*kernel_trace.csv" | head -1)
  echo "== streams $S: $(python -c "import json; b=json.load(open('$out/bench_s$S.json')); print(round(b['value']), 'MSAs/s', b['ms_per_step'], 'ms/step')")"
  python tools/trace_overlap.py $f 0.4
  rm -rf $out/trace_s$S
done

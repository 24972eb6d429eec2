#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r03_c45; mkdir -p $out
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 4 --warmup 2 --workers 0"
for s in 1 4; do
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_s$s -- python3 bench.py $o --streams $s > $out/bench_s$s.json 2> $out/err_s$s.txt
  f=$(find $out/trace_s$s -name "*kernel_trace.csv" | head -1)
  echo "== in-process, $s stream(s): $(python -c "import json;b=json.load(open('$out/bench_s$s.json'));print(round(b['value']),'MSAs/s',b['ms_per_step'],'ms/step')")"
  python tools/trace_concurrency.py $f 2 6 $s | tee $out/concurrency_s$s.txt
  gzip -c $f > $out/kernel_trace_s$s.csv.gz; rm -rf $out/trace_s$s
done

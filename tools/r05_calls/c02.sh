#!/bin/bash
# round 5, call 2: kernel timeline of a 3 750-alignment shard (in-process bench under rocprofv3): where the 52 ms go
out=gpurun_out/r05_c02; mkdir -p $out
export TMPDIR=/tmp
for mode in planned first; do
  fp=""; [ $mode = first ] && fp="--first-pass"
  MPRG_KM_SIDE_STREAMS=1 rocprofv3 --kernel-trace --output-format csv -d $out/tr_$mode -- python3 bench.py --workers 0 --streams 2 --batch 3750 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $fp > $out/bench_$mode.json 2> $out/bench_$mode.err
  f=$(find $out/tr_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_overlap.py $f 0.5 > $out/overlap_$mode.txt 2>&1
  gzip -c $f > $out/kernel_trace_$mode.csv.gz
  rm -rf $out/tr_$mode
  cat $out/overlap_$mode.txt
  tail -c 600 $out/bench_$mode.json | head -c 300; echo
done
ls -la $out

#!/bin/bash
# round 5, call 9: multi-rank command line timings (gloo, one GPU) + stage trace of the one-GPU command line
out=gpurun_out/r05_c09; mkdir -p $out
timeout 1500 python tools/multi_rank_timing.py 10000 1,2,4 $out/multi_rank_10000.json > $out/multi_rank.txt 2>&1; cut -c1-420 $out/multi_rank.txt
timeout 900 python tools/cli_bench.py 30000 16 p:MPRG_PIPELINE_TRACE=1 a:MPRG_PIPELINE_TRACE=1 p a > $out/cli_trace.txt 2>&1
grep -v "^\[pipeline\] chunk" $out/cli_trace.txt | head -30
grep "chunk 1[0-2]:" $out/cli_trace.txt | head -40

#!/bin/bash
# round 5, call 18: time line of the command line (-O p), every stage stamped
out=gpurun_out/r05_c18; mkdir -p $out
timeout 900 python tools/cli_bench.py 30000 16 p:MPRG_PIPELINE_TRACE=1 > $out/cli_trace.txt 2>&1
grep -E "^-O|starts|ready|closed|exits" $out/cli_trace.txt
grep -E "chunk (8|9|10|11):" $out/cli_trace.txt | head -60

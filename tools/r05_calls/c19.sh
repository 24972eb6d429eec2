#!/bin/bash
# round 5, call 19: the command line with two interleaved engines, parser state released off the build thread, pwritev batching
out=gpurun_out/r05_c19; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_cli.py -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 1500 python tools/cli_bench.py 30000 16 p a p:MPRG_PIPELINE_ENGINES=1 a:MPRG_PIPELINE_ENGINES=1 p a p:MPRG_WRITE_THREADS=4 a:MPRG_WRITE_THREADS=4 p:MPRG_WRITE_THREADS=1 p:MPRG_PIPELINE_TRACE=1 > $out/cli.txt 2>&1
grep -E "^-O|closed" $out/cli.txt
grep -E "chunk (9|10):" $out/cli.txt | tail -16

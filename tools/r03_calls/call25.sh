#!/bin/bash
# CLI after the folding CRC-32 + pooled one-pass encoders: write threads per container 2 / 1 / 8, chunk sizes
mkdir -p gpurun_out/r03_c25
MPRG_PIPELINE_TRACE=1 python tools/cli_bench.py 30000 16 a a:MPRG_WRITE_THREADS=1 a:MPRG_WRITE_THREADS=8 a:MPRG_CHUNK=2048 a:MPRG_CHUNK=8192 > gpurun_out/r03_c25/cli.txt 2>&1
grep -v "chunk" gpurun_out/r03_c25/cli.txt
python -m pytest tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -2

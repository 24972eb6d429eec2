#!/bin/bash
# the torch-free runtime backend: parity file on both backends, CLI tests, CLI rate (default = runtime backend, fast exit)
mkdir -p gpurun_out/r03_c27
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -5
MPRG_PIPELINE_TRACE=1 python tools/cli_bench.py 30000 16 a a:MPRG_BACKEND=torch a a:MPRG_CHUNK=1536 a:MPRG_CHUNK=1024 a:MPRG_CHUNK=3072 > gpurun_out/r03_c27/cli.txt 2>&1
grep -v "chunk" gpurun_out/r03_c27/cli.txt | grep -v "^   "

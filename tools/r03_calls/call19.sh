#!/bin/bash
mkdir -p gpurun_out/r03_c19
MPRG_PIPELINE_TRACE=1 python tools/cli_bench.py 30000 16 a bg > gpurun_out/r03_c19/cli_t16.txt 2>&1
grep -v "^\[pipeline\]" gpurun_out/r03_c19/cli_t16.txt; grep "chunk [34]:" gpurun_out/r03_c19/cli_t16.txt | head -16
python -m pytest tests/test_gpu_cli.py tests/test_gpu_update.py -m gpu -x -q 2>&1 | tail -2

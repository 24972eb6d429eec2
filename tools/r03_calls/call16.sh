#!/bin/bash
mkdir -p gpurun_out/r03_c16
MPRG_PIPELINE_TRACE=1 python tools/cli_bench.py 12000 16 a > gpurun_out/r03_c16/cli_t16.txt 2>&1
cat gpurun_out/r03_c16/cli_t16.txt

#!/bin/bash
mkdir -p gpurun_out/r03_c17
MPRG_PIPELINE_TRACE=1 python tools/cli_bench.py 30000 16 a bg > gpurun_out/r03_c17/cli_t16.txt 2>&1
grep -v "^\[pipeline\]" gpurun_out/r03_c17/cli_t16.txt; grep "chunk [34]:" gpurun_out/r03_c17/cli_t16.txt | head -24
python tools/cli_bench.py 30000 32 a > gpurun_out/r03_c17/cli_t32.txt 2>&1; cat gpurun_out/r03_c17/cli_t32.txt

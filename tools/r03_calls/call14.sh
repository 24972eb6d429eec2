#!/bin/bash
mkdir -p gpurun_out/r03_c14
python tools/cli_bench.py 10000 4 a bg > gpurun_out/r03_c14/cli_t4.txt 2>&1
python tools/cli_bench.py 10000 16 a bg > gpurun_out/r03_c14/cli_t16.txt 2>&1
cat gpurun_out/r03_c14/cli_t4.txt gpurun_out/r03_c14/cli_t16.txt

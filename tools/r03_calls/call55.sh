#!/bin/bash
# the command-line leg five times on the final tree (one set of 30 000 files)
mkdir -p gpurun_out/r03_c55
python tools/cli_bench.py 30000 16 a a a a a > gpurun_out/r03_c55/cli_five_runs.txt 2>&1
grep "files in" gpurun_out/r03_c55/cli_five_runs.txt

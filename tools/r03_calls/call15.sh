#!/bin/bash
mkdir -p gpurun_out/r03_c15
python -m pytest tests/test_gpu_cli.py tests/test_gpu_update.py -m gpu -x -q > gpurun_out/r03_c15/pytest.txt 2>&1; tail -3 gpurun_out/r03_c15/pytest.txt
python tools/cli_bench.py 10000 16 a bg > gpurun_out/r03_c15/cli_t16.txt 2>&1
cat gpurun_out/r03_c15/cli_t16.txt
df -h /tmp | tail -1; nproc

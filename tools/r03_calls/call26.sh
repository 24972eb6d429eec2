#!/bin/bash
mkdir -p gpurun_out/r03_c26
python tools/startup_probe.py > gpurun_out/r03_c26/startup.txt 2>&1
cat gpurun_out/r03_c26/startup.txt
df -h /tmp /dev/shm | cat; mount | grep -E " /tmp | / " | head -3
MPRG_PIPELINE_TRACE=1 python tools/cli_bench.py 30000 16 a:MPRG_CHUNK=2048 a:MPRG_CHUNK=2048,MPRG_FAST_EXIT=1 a:MPRG_CHUNK=1024,MPRG_FAST_EXIT=1 a:MPRG_CHUNK=1536,MPRG_FAST_EXIT=1 a:MPRG_CHUNK=3072,MPRG_FAST_EXIT=1 > gpurun_out/r03_c26/cli.txt 2>&1
grep -v "chunk" gpurun_out/r03_c26/cli.txt | grep -v "^   "

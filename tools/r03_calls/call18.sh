#!/bin/bash
mkdir -p gpurun_out/r03_c18
python tools/write_probe.py /tmp > gpurun_out/r03_c18/write_probe_tmp.txt 2>&1; cat gpurun_out/r03_c18/write_probe_tmp.txt
python tools/write_probe.py /dev/shm > gpurun_out/r03_c18/write_probe_shm.txt 2>&1; cat gpurun_out/r03_c18/write_probe_shm.txt
MPRG_WRITERS=1 python tools/cli_bench.py 30000 16 a > gpurun_out/r03_c18/cli_w1.txt 2>&1; cat gpurun_out/r03_c18/cli_w1.txt
MPRG_WRITERS=2 python tools/cli_bench.py 30000 16 a > gpurun_out/r03_c18/cli_w2.txt 2>&1; cat gpurun_out/r03_c18/cli_w2.txt

#!/bin/bash
mkdir -p gpurun_out/r03_c28
python tools/mmap_write_probe.py > gpurun_out/r03_c28/mmap_probe.txt 2>&1
cat gpurun_out/r03_c28/mmap_probe.txt

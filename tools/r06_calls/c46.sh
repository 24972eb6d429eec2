#!/bin/bash
# round 6, call 46: the whole GPU suite on the round's last sources
out=gpurun_out/r06_c46; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt

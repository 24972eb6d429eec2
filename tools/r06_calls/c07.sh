#!/bin/bash
# round 6, call 7: the whole GPU suite on the LDS form as the default (with the one-rank RCCL tests of the command line and of bench.py)
out=gpurun_out/r06_c07; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -15 $out/pytest_gpu.txt

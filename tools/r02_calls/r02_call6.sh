#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02f
mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
tail -6 $out/pytest_gpu.txt
bash tools/measure_round.sh r02f

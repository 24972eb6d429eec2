#!/bin/bash
# round 6, call 15: the whole GPU suite (new: earlier KMeans forms, the diagnostic build's fused loops, planned forests with the side stream,
# one-rank RCCL) and the round's measurement set on the last sources
out=gpurun_out/r06_c15; mkdir -p $out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; tail -4 $out/pytest_gpu.txt
bash tools/measure_round.sh r06 > $out/measure.txt 2>&1; tail -5 $out/measure.txt | cut -c1-300

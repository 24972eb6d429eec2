#!/bin/bash
# GPU-box helper: GPU test suite + the round's measurement set on the same tree
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02/pytest_gpu.txt
bash tools/measure_round.sh r02
[ -f make_prg_amd/_lib/libmprg_hip_timing.so ] && python tools/phase_timing.py 2048 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/kmeans_phase_cycles_final.txt || true

#!/bin/bash
# GPU-box helper: parity sweeps outside the test suite on the round's final kernels
export TMPDIR=/tmp
out=gpurun_out/r02ai
mkdir -p $out
timeout 900 python tools/parity_sweep_nasty.py 6000 2>&1 | grep -v amdgpu.ids | tee $out/sweep_nasty_small.txt | tail -8
timeout 1200 python tools/parity_sweep_nasty.py 6000 medium 2>&1 | grep -v amdgpu.ids | tee $out/sweep_nasty_medium.txt | tail -8
timeout 1200 python tools/parity_sweep.py 8000 500000 2>&1 | grep -v amdgpu.ids | tee $out/sweep_config_c.txt | tail -4

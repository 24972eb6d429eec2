#!/bin/bash
# parity sweeps on the final kernels of round 3 (group loads in the KMeans kernels, device-resident forest): HIP path vs oracle
out=gpurun_out/r03_c53; mkdir -p $out
timeout 900 python tools/parity_sweep_nasty.py 6000 2>&1 | grep -v amdgpu.ids | tee $out/sweep_nasty_small.txt | tail -8
timeout 1200 python tools/parity_sweep_nasty.py 6000 medium 2>&1 | grep -v amdgpu.ids | tee $out/sweep_nasty_medium.txt | tail -8
timeout 1500 python tools/parity_sweep.py 12000 700000 2>&1 | grep -v amdgpu.ids | tee $out/sweep_config_c.txt | tail -4

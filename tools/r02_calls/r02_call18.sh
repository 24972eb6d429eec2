#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02r
mkdir -p $out
python tools/contention_probe.py 1 3000 2>&1 | grep -v amdgpu.ids
python tools/contention_probe.py 10 3000 2>&1 | grep -v amdgpu.ids
GPU_MAX_HW_QUEUES=1 python tools/contention_probe.py 10 3000 2>&1 | grep -v amdgpu.ids
GPU_MAX_HW_QUEUES=2 python tools/contention_probe.py 10 3000 2>&1 | grep -v amdgpu.ids
HSA_ENABLE_SDMA=0 python tools/contention_probe.py 10 3000 2>&1 | grep -v amdgpu.ids

#!/bin/bash
export TMPDIR=/tmp
export MPRG_HIP_LIB=$PWD/make_prg_amd/_lib/variants/libmprg_hip_r20.so
for ni in 10 20 5; do python tools/ninit_probe.py $ni 3000 2>&1 | grep -v amdgpu.ids | tail -2; done

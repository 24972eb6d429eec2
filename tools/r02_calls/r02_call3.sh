#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02c
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -DKM_PHASE_TIMING make_prg_amd/csrc/mprg_api.hip -o make_prg_amd/_lib/libmprg_hip_timing.so
echo "== LDS path" > $out/phase.txt
python tools/phase_timing.py 2048 >> $out/phase.txt 2>&1
echo "== global path" >> $out/phase.txt
MPRG_KMEANS_LDS=0 python tools/phase_timing.py 2048 >> $out/phase.txt 2>&1
cat $out/phase.txt

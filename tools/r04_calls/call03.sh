#!/bin/bash
# phase cycles of a fit inside the fused loop against the per-round kernels (diagnostic build)
out=gpurun_out/r04_c03; mkdir -p $out
for mode in rounds fused; do
  MPRG_KLOOP=$mode python tools/phase_timing.py 4096 > $out/phase_$mode.txt 2>&1
  echo "== $mode"; grep -v "k_partition" $out/phase_$mode.txt | cut -c1-120
done

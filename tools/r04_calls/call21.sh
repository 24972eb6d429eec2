#!/bin/bash
# workgroup sizes of the per-view kernels (most views below the root are a few hundred cells): partition fused form, dedupe
out=gpurun_out/r04_c21; mkdir -p $out
for pf in 256 128 64; do for dd in 256 128 64; do
  MPRG_PF_THREADS=$pf MPRG_DD_THREADS=$dd MPRG_BACKEND=runtime MPRG_SPECULATIVE=0 python tools/forest_profile.py 7500 3 > $out/f_${pf}_${dd}.txt 2>&1
  echo "pf $pf dd $dd: $(grep -E 'mprg_partition|mprg_ungap_dedupe' $out/f_${pf}_${dd}.txt | awk '{print $1, $2}' | tr '\n' ' ')"
done; done

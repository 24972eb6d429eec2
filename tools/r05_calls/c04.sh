#!/bin/bash
# round 5, call 4: do forests on streams of one process overlap, with more hardware queues?  host cost of enqueuing a forest
out=gpurun_out/r05_c04; mkdir -p $out
for q in 4 8 16; do for s in 1 2 4 8; do
  GPU_MAX_HW_QUEUES=$q MPRG_KM_SIDE_STREAMS=0 timeout 300 python tools/overlap_probe.py 3750 $s 2>&1 | tail -1 | sed "s/^/q$q noside: /"
done; done | tee $out/overlap.txt
for q in 8 16; do for s in 1 2 4; do
  GPU_MAX_HW_QUEUES=$q MPRG_KM_SIDE_STREAMS=1 timeout 300 python tools/overlap_probe.py 3750 $s 2>&1 | tail -1 | sed "s/^/q$q side: /"
done; done | tee -a $out/overlap.txt

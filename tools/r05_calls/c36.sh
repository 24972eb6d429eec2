#!/bin/bash
# round 5, call 36: where a wide fit's cycles go (diagnostic build, thread 0's shader clock between barriers) on deep alignments
out=gpurun_out/r05_c36; mkdir -p $out
for sz in "2000 4000" "10000 20000"; do
  timeout 900 python tools/phase_timing.py deep $sz > $out/phase_deep_${sz% *}.txt 2>&1; echo deep $sz; grep -v "k_partition\|^loop" $out/phase_deep_${sz% *}.txt | head -20
done

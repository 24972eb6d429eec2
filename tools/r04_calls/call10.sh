#!/bin/bash
# one worker process, several host threads / streams (sub-batches): how much of the multi-process gain do threads reach?
out=gpurun_out/r04_c10; mkdir -p $out
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 8 --warmup 2"
run() {  # tag, batch, workers, streams, env...
  tag=$1; B=$2; W=$3; S=$4; shift 4
  env "$@" python bench.py $o --batch $B --workers $W --streams $S > $out/bench_$tag.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_$tag.json"))
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step verified", b["config"]["verified"]["mismatches"])
P
}
for B in 3750 30000; do for S in 1 2 4; do
run fused_${B}_w1_s$S $B 1 $S MPRG_KLOOP=fused
done; done
run fused_3750_w2_s2 3750 2 2 MPRG_KLOOP=fused
run fused_30000_w2_s2 30000 2 2 MPRG_KLOOP=fused
run rounds_3750_w1_s4 3750 1 4 MPRG_KLOOP=rounds
run rounds_30000_w1_s4 30000 1 4 MPRG_KLOOP=rounds

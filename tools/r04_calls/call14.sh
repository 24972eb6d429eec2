#!/bin/bash
# worker-count sweep, fused (speculative) against rounds (per-step host), 30 000 alignments per step
out=gpurun_out/r04_c14; mkdir -p $out
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 8 --warmup 2"
run() {  # tag, batch, workers, streams, env...
  tag=$1; B=$2; W=$3; S=$4; shift 4
  env "$@" python bench.py $o --batch $B --workers $W --streams $S > $out/bench_$tag.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_$tag.json"))
c=b["config"]
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; workers", c["host_worker_processes_per_gpu"], "waits/step", c["host_waits_per_step"], "verified", c["verified"]["mismatches"])
P
}
for W in 2 3 6 8; do run fused_w$W 30000 $W 1 MPRG_KLOOP=fused; done
for W in 2 6 8; do run rounds_w$W 30000 $W 1 MPRG_KLOOP=rounds; done

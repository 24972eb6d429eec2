#!/bin/bash
# workers x engines per worker, 30 000 and 3 750 alignments per step: which shape is the default
out=gpurun_out/r04_c19; mkdir -p $out
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --steps 8 --warmup 2"
run() {  # tag, batch, workers, streams, env...
  tag=$1; B=$2; W=$3; S=$4; shift 4
  env "$@" python bench.py $o --batch $B --workers $W --streams $S > $out/bench_$tag.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_$tag.json"))
c=b["config"]
print("$tag:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; waits/step", c["host_waits_per_step"], "verified", c["verified"]["mismatches"])
P
}
run 30000_w4_s1_auto 30000 4 1
run 30000_w4_s2_fused 30000 4 2 MPRG_KLOOP=fused
run 30000_w2_s2_fused 30000 2 2 MPRG_KLOOP=fused MPRG_KM_SIDE_STREAMS=1
run 30000_w1_s2_fused 30000 1 2 MPRG_KLOOP=fused
run 30000_w1_s4_fused 30000 1 4 MPRG_KLOOP=fused
run 30000_w2_s4_fused 30000 2 4 MPRG_KLOOP=fused
run 3750_w2_s2 3750 2 2 MPRG_KM_SIDE_STREAMS=1
run 3750_w4_s2 3750 4 2 MPRG_KM_SIDE_STREAMS=1
run 3750_w2_s1 3750 2 1 MPRG_KM_SIDE_STREAMS=1

#!/bin/bash
# one host thread feeding several engines (streams) without waits: sub-batches per worker process
out=gpurun_out/r04_c12; mkdir -p $out
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 8 --warmup 2"
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
for B in 3750 30000; do
for S in 1 2 3 4 6; do run ${B}_w1_s$S $B 1 $S; done
run ${B}_w2_s2 $B 2 2
run ${B}_w2_s3 $B 2 3
run ${B}_w4_s1 $B 4 1
done

#!/bin/bash
# the loop's general and small forms side by side (side stream inside mprg_forest_level): one worker, small shards; streams 1 / 2
out=gpurun_out/r04_c18; mkdir -p $out
python -m pytest tests/test_gpu_speculative.py -m gpu -x -q 2>&1 | tail -2
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
for B in 3750 7500; do
run ${B}_side0_s1 $B 1 1 MPRG_KM_SIDE_STREAMS=0 MPRG_KLOOP=fused
run ${B}_side1_s1 $B 1 1 MPRG_KM_SIDE_STREAMS=1 MPRG_KLOOP=fused
run ${B}_side0_s2 $B 1 2 MPRG_KM_SIDE_STREAMS=0 MPRG_KLOOP=fused
run ${B}_side1_s2 $B 1 2 MPRG_KM_SIDE_STREAMS=1 MPRG_KLOOP=fused
done

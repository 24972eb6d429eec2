#!/bin/bash
# fused clustering loop (mprg_cluster_loop) against the per-round launches: GPU parity, then device time / wall of one worker
# at 3 750 / 7 500 / 30 000 alignments per step
out=gpurun_out/r04_c01; mkdir -p $out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $out/pytest_gpu.txt
for b in 3750 7500; do for mode in fused rounds; do
  MPRG_KLOOP=$mode MPRG_BACKEND=runtime python tools/forest_profile.py $b 4 > $out/forest_${b}_$mode.txt 2>&1
  echo "== $b $mode"; grep -E "^step 3|device time|cluster_loop|kmeans_fit|cluster_further|kloop" $out/forest_${b}_$mode.txt | cut -c1-150
done; done
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 6 --warmup 2"
for b in 3750 30000; do for mode in fused rounds; do for w in 1 4; do
  MPRG_KLOOP=$mode python bench.py $o --batch $b --workers $w > $out/bench_${b}_${mode}_w$w.json 2> $out/err.txt || tail -5 $out/err.txt
  python - <<P
import json
b=json.load(open("$out/bench_${b}_${mode}_w$w.json"))
x=b["roofline"]["exclusive_pass"]
print("$b $mode w$w:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step; excl wall", x["wall_ms"], "device", x["device_ms"], "launches", x["launches"], "waits", x["host_waits"], "verified", b["config"]["verified"]["mismatches"])
P
done; done; done

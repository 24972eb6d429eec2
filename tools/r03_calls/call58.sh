#!/bin/bash
# split form for under-filled KMeans launches: threshold sweep at several shard sizes (one worker and four), parity on the GPU
mkdir -p gpurun_out/r03_c58
MPRG_KM_SPLIT_BELOW=100000 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "synthetic or integration or known_answers or side" 2>&1 | tail -2
o="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 12 --warmup 3"
for cfg in "3750 1" "7500 1" "30000 1" "30000 4" "3750 4"; do set -- $cfg; for t in 0 100 300 1000 3000; do
  MPRG_KM_SPLIT_BELOW=$t python bench.py $o --batch $1 --workers $2 > gpurun_out/r03_c58/b_$1_w$2_t$t.json 2> gpurun_out/r03_c58/err.txt
  python - <<P
import json
b=json.load(open("gpurun_out/r03_c58/b_$1_w$2_t$t.json"))
print("batch $1 workers $2 split<=$t:", round(b["value"]), "MSAs/s", b["ms_per_step"], "ms/step", b["config"]["verified"]["mismatches"])
P
done; done | tee gpurun_out/r03_c58/summary.txt

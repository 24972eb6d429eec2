#!/bin/bash
# round 5, call 26: the k = 7..10 small KMeans kernel at eight workgroups per CU (one relocation stack, shorter pair list): parity + times + rates
out=gpurun_out/r05_c26; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_7500.txt 2>&1
grep -E "per launch mprg_kmeans_fit_small|device time|mprg_kmeans_fit" $out/profile_7500.txt | cut -c1-300
for rep in 1 2; do
timeout 500 python bench.py --batch 30000 --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg > $out/p30000_$rep.json 2> $out/p30000_$rep.err
python - <<PY
import json
d=json.loads(open("$out/p30000_$rep.json").read().strip().splitlines()[-1]); print("30000:", d["value"], d["ms_per_step"], d["config"]["verified"]["mismatches"], [(k["entry_point"], k["ms"]) for k in d["roofline"]["kernels"][:3]])
PY
done

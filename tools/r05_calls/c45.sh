#!/bin/bash
# round 5, call 45: K6's LDS form by a thread per sample pair (all chains of the pair together): parity, per-launch times of mprg_kmeans_prepare
# at config C, bench value; the deep alignments once more (tiled tables with the pair state as local variables)
out=gpurun_out/r05_c45; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddeep.py -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_7500.txt 2>&1
grep -E "per launch mprg_kmeans_prepare|device time|mprg_kmeans_(prepare|fit|fit_small) " $out/profile_7500.txt | cut -c1-200
for rep in 1 2; do
  timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg > $out/bench_$rep.json 2> $out/bench_$rep.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$rep.json").read().strip().splitlines()[-1]); c=d["config"]
print("bench $rep", d["value"], d["ms_per_step"], "bad", c["verified"]["mismatches"])
PY
done
for sz in "2000 4000" "10000 20000"; do
  MPRG_BACKEND=runtime timeout 900 python tools/deep_profile.py $sz --passes 2 > $out/deep_${sz% *}.txt 2>&1
  grep -E "prg_sha256" $out/deep_${sz% *}.txt | tail -1 | cut -c1-150; grep -E "mprg_kmeans_(fit_wide|prepare_big) " $out/deep_${sz% *}.txt | head -2
done

#!/bin/bash
# round 5, call 33: the small KMeans form by ONE wavefront per fit (64 threads, MPRG_KMS_THREADS) beside two: per-launch times and bench value on one box
out=gpurun_out/r05_c33; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
for t in 128 64; do
  MPRG_KMS_THREADS=$t MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_kms$t.txt 2>&1
  echo threads $t; grep -E "per launch mprg_kmeans_fit_small|per launch mprg_partition|device time|mprg_kmeans_fit_small |mprg_kmeans_fit |mprg_partition " $out/profile_kms$t.txt | cut -c1-220
done
run() {
  label=$1; shift
  env "$@" timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
for rep in 1 2; do run kms128_$rep MPRG_KMS_THREADS=128; run kms64_$rep MPRG_KMS_THREADS=64; done

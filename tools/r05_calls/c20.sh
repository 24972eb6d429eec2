#!/bin/bash
# round 5, call 20: mprg_cluster_further in one workgroup for problems that fit LDS: parity + per-entry-point times (7 500, exclusive) + rates
out=gpurun_out/r05_c20; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
for one in 1 0; do
  MPRG_CF_ONE=$one MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_cf$one.txt 2>&1
  grep -E "per launch mprg_cluster_further|device time|mprg_cluster_further " $out/profile_cf$one.txt | cut -c1-300
done
run() {
  label=$1; shift
  env "$@" timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "misses", c["plan_misses_per_step"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
ARGS="--batch 30000" run p30000_cf1 MPRG_CF_ONE=1
ARGS="--batch 30000" run p30000_cf0 MPRG_CF_ONE=0
ARGS="--batch 3750 --first-pass" run f3750_cf1 MPRG_CF_ONE=1
ARGS="--batch 3750 --first-pass" run f3750_cf0 MPRG_CF_ONE=0
ARGS="--batch 3750" run p3750_cf1 MPRG_CF_ONE=1

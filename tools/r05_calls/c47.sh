#!/bin/bash
# round 5, call 47: mid-sized views' ungap + hashes + row groups by one workgroup and launch (k_dedupe_view; MPRG_DD_VIEW=0: three launches)
# and the vectorized capacity prediction: parity, per-launch times, bench value and the 3 750-alignment first pass, with / without
out=gpurun_out/r05_c47; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_speculative.py -x -q > $out/pytest.txt 2>&1; tail -2 $out/pytest.txt
for v in 1 0; do
  MPRG_DD_VIEW=$v MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_ddview$v.txt 2>&1
  echo dd_view $v; grep -E "per launch mprg_ungap_dedupe|device time|mprg_ungap_dedupe " $out/profile_ddview$v.txt | cut -c1-200
done
run() {
  label=$1; shift
  env "$@" timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "misses", c.get("plan_misses_per_step"), "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
for rep in 1 2; do
  ARGS="" run v1_$rep MPRG_DD_VIEW=1
  ARGS="" run v0_$rep MPRG_DD_VIEW=0
  ARGS="--batch 3750 --first-pass" run f3750_v1_$rep MPRG_DD_VIEW=1
  ARGS="--batch 3750 --first-pass" run f3750_v0_$rep MPRG_DD_VIEW=0
  ARGS="--batch 3750" run p3750_v1_$rep MPRG_DD_VIEW=1
done

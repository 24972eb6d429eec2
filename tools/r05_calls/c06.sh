#!/bin/bash
# round 5, call 6: small views by a wavefront each (k_partition_wave, k_dedupe_wave): parity + rates with and without
out=gpurun_out/r05_c06; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt; tail -3 $out/pytest.txt
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg $ARGS > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]; r=d["roofline"]
    print("$label", d["value"], d["ms_per_step"], "waits", c["host_waits_per_step"], "misses", c["plan_misses_per_step"], "bad", c["verified"]["mismatches"], "excl dev ms", r["exclusive_pass"]["device_ms"])
    print("     ", [(k["entry_point"].replace("mprg_",""), k["ms"]) for k in r["kernels"]])
except Exception as e: print("$label failed", e)
PY
}
for wv in 1 0; do
ARGS="--batch 30000" run p30000_wv$wv MPRG_WAVE_VIEWS=$wv
ARGS="--batch 3750" run p3750_wv$wv MPRG_WAVE_VIEWS=$wv
ARGS="--batch 3750 --first-pass" run f3750_wv$wv MPRG_WAVE_VIEWS=$wv
done

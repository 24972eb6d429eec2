#!/bin/bash
# GPU-box helper: restarts of the biggest fits of a round in workgroups of their own (MPRG_KM_SPLIT_WORK = D V k threshold)
export TMPDIR=/tmp
out=gpurun_out/r02u
mkdir -p $out
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
for t in 0 200000 60000 20000 8000; do
  MPRG_KM_SPLIT_WORK=$t timeout 600 python bench.py $inproc > $out/split_$t.json 2> $out/split_$t.err
  python - $out/split_$t.json $t <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("split", sys.argv[2], "device_ms", r["exclusive_pass"]["device_ms"], "launches", r["exclusive_pass"]["launches"], [(k["entry_point"], k["ms"], k["launches"]) for k in r["kernels"][:5]], "verified", d["config"]["verified"]["mismatches"])
PY
done
for t in 0 20000; do
MPRG_KM_SPLIT_WORK=$t python bench.py --no-cpu-baseline --no-end-to-end > $out/b_$t.json 2> $out/b_$t.err; cut -c1-170 $out/b_$t.json
done

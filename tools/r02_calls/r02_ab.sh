#!/bin/bash
# A/B on one box: default bench (10 workers, 30k) with the two-launch KMeans form and with persistent workgroups
export TMPDIR=/tmp
out=gpurun_out/r02h
mkdir -p $out
for rep in 1 2; do
MPRG_KMEANS_SLOTS=0 python bench.py --no-cpu-baseline --no-end-to-end --steps 3 > $out/ab_two_launch_$rep.json 2> $out/ab_two_launch_$rep.err
python bench.py --no-cpu-baseline --no-end-to-end --steps 3 > $out/ab_persistent_$rep.json 2> $out/ab_persistent_$rep.err
done
python - <<'PY'
import json
for n in ("two_launch_1", "persistent_1", "two_launch_2", "persistent_2"):
    d = json.loads(open(f"gpurun_out/r02h/ab_{n}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(n, d["value"], "MSAs/s", d["ms_per_step"], "ms/step; exclusive:", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"]) for k in r["kernels"][:3]])
PY

#!/bin/bash
export TMPDIR=/tmp
bash tools/measure_round.sh r02
out=gpurun_out/r02
# the persistent form, same box, for profiles/r02/kmeans_forms.md
MPRG_KMEANS_SLOTS=1 python bench.py --no-cpu-baseline --no-end-to-end --steps 3 > $out/bench_persistent.json 2> $out/bench_persistent.err
python bench.py --no-cpu-baseline --no-end-to-end --steps 3 > $out/bench_two_launch_again.json 2> $out/bench_two_launch_again.err
python - <<'PY'
import json
for n in ("bench_default", "bench_persistent", "bench_two_launch_again"):
    d = json.loads(open(f"gpurun_out/r02/{n}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(n, d["value"], "MSAs/s; exclusive:", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"]) for k in r["kernels"][:6]])
PY
cat $out/pmc_summary.txt | head -60

#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02q
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.txt
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
timeout 600 python bench.py $inproc > $out/inproc.json 2> $out/inproc.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02q/inproc.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"], k["frac"]) for k in r["kernels"][:8]], "frac", r["frac"], "verified", d["config"]["verified"]["mismatches"])
PY
python bench.py --no-cpu-baseline --no-end-to-end > $out/bench10.json 2> $out/bench10.err; cut -c1-200 $out/bench10.json
python bench.py --no-cpu-baseline --no-end-to-end > $out/bench10b.json 2> $out/bench10b.err; cut -c1-200 $out/bench10b.json

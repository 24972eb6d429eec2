#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02as
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest_gpu.txt
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
timeout 600 python bench.py $inproc > $out/inproc.json 2> $out/inproc.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02as/inproc.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"]) for k in r["kernels"][:7]], "frac", r["frac"], "verified", d["config"]["verified"]["mismatches"])
PY
python tools/phase_timing.py 2048 2>&1 | grep -v amdgpu.ids | head -11
for rep in 1 2; do python bench.py --no-cpu-baseline --no-end-to-end > $out/b$rep.json 2> $out/b$rep.err; cut -c1-170 $out/b$rep.json; done

#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02d
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_update.py tests/test_gpu_cli.py tests/test_gpu_parity.py -x -q > $out/pytest_update.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_update.txt
tail -4 $out/pytest_update.txt
timeout 900 python bench.py --batch 3000 --workers 4 --steps 2 --cpu-sample 512 > $out/bench_small.json 2> $out/bench_small.err
echo "bench rc=$?"
tail -3 $out/bench_small.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02d/bench_small.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "cpu_baseline")}, indent=0))
print(json.dumps(d["config"]["verified"]), json.dumps(d["config"]["end_to_end"], indent=0))
r = d["roofline"]
print({k: r[k] for k in r if k != "kernels"})
for k in r["kernels"]: print(k)
PY

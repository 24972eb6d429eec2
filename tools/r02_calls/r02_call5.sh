#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02e
mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_update.py -x -q > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/pytest.txt; tail -3 $out/pytest.txt
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?"; tail -2 $out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02e/bench_default.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step")}), d["cpu_baseline"]["value"], d["cpu_baseline"]["end_to_end_value"])
print(json.dumps(d["config"]["verified"]), json.dumps(d["config"]["end_to_end"]))
r = d["roofline"]
print({k: r[k] for k in r if k != "kernels"})
for k in r["kernels"]: print(k)
PY

#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02g
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ddeep.py tests/test_gpu_cli.py tests/test_gpu_update.py -x -q > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
tail -4 $out/pytest_gpu.txt
one="--workers 1 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 2"
MPRG_KMEANS_SLOTS=0 python bench.py $one > $out/bench_one_two_launch.json 2> $out/bench_one_two_launch.err
python bench.py $one > $out/bench_one_persistent.json 2> $out/bench_one_persistent.err
python - <<'PY'
import json
for n in ("two_launch", "persistent"):
    try:
        d = json.loads(open(f"gpurun_out/r02g/bench_one_{n}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(n, d["value"], "MSAs/s", r["exclusive_pass"], d["config"]["verified"]["mismatches"])
        for k in r["kernels"]: print("   ", k["entry_point"], k["ms"], k["launches"], k["achieved_GBps"])
    except Exception as e:
        print(n, "failed", e)
PY
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err
echo "bench rc=$?"; tail -2 $out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02g/bench_default.json").read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step")}), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print(json.dumps(d["config"]["verified"]), json.dumps(d["config"]["end_to_end"]))
r = d["roofline"]
print({k: r[k] for k in r if k != "kernels"})
for k in r["kernels"]: print(k)
PY

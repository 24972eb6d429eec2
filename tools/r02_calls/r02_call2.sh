#!/bin/bash
# round-2 GPU call 2: GPU test-suite with the LDS-resident KMeans, then one-stream exclusive timings (old vs new path)
export TMPDIR=/tmp
out=gpurun_out/r02b
mkdir -p $out
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
tail -5 $out/pytest_gpu.txt
one="--workers 0 --streams 1 --batch 8192 --no-cpu-baseline --steps 2"
MPRG_KMEANS_LDS=0 python bench.py $one > $out/bench_one_stream_global.json 2> $out/bench_one_stream_global.err
python bench.py $one > $out/bench_one_stream_lds.json 2> $out/bench_one_stream_lds.err
python - <<'PY'
import json
for n in ("global", "lds"):
    try:
        d = json.loads(open(f"gpurun_out/r02b/bench_one_stream_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], "MSAs/s  device ms/step", d["config"]["device_ms_per_step"])
        for k, v in d["config"]["kernels"].items():
            print("   ", k, v)
    except Exception as e:
        print(n, "failed", e)
PY
python bench.py --steps 2 > $out/bench_default.json 2> $out/bench_default.err
cut -c1-400 $out/bench_default.json

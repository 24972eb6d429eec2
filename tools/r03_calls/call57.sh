#!/bin/bash
mkdir -p gpurun_out/r03_c57
python bench.py > gpurun_out/r03_c57/bench_default.json 2> gpurun_out/r03_c57/bench_default.err; echo "rc=$?"
python - <<'P'
import json
lines=[l for l in open("gpurun_out/r03_c57/bench_default.json").read().splitlines() if l.strip()]
print("lines", len(lines))
b=json.loads(lines[-1])
r=b["roofline"]
print({k:b[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data")})
print("roofline", {k:r[k] for k in ("bound","achieved","peak","unit","frac","traffic","kernel")})
print("traffic_note", r["traffic_note"])
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"], b["cpu_baseline"]["kind"])
c=b["config"]; print(c["single_worker"]["value"], c["single_worker"]["fraction_of_value"], c["cli"]["value"], c["end_to_end"]["value"], c["verified"]["mismatches"])
P

#!/bin/bash
# full GPU test suite + the default bench line (shard projection, -O p CLI leg, reference tie)
out=gpurun_out/r04_c17; mkdir -p $out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $out/pytest_gpu.txt
( time python bench.py > $out/bench_default.json 2> $out/bench_default.err ) 2>&1 | tail -3
python - <<P
import json
b=json.load(open("$out/bench_default.json"))
c=b["config"]
print("value", b["value"], "ms/step", b["ms_per_step"], "verified", c["verified"]["mismatches"], "waits/step", c["host_waits_per_step"], "calls/step", c["launches_per_step"])
print("single_worker", c["single_worker"] and {k: c["single_worker"].get(k) for k in ("value","fraction_of_value")})
print("cli", c["cli"] and {k: c["cli"].get(k) for k in ("value","seconds","error")}, c["cli"] and c["cli"].get("prg_only"))
for s in (c["shard_projection"] or {}).get("shards", []): print(s)
print("roofline", {k: b["roofline"][k] for k in ("entry_point","achieved","frac","traffic")})
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["sample"][:80])
print("tie", c["verified"].get("reference_tie", {}).get("stable_loci"), c["verified"].get("reference_tie", {}).get("unstable_loci"))
P
tail -3 $out/bench_default.err

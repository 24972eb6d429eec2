#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02al
mkdir -p $out
run() { tag=$1; shift; python bench.py --no-cpu-baseline --no-end-to-end "$@" > $out/$tag.json 2> $out/$tag.err; python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "workers", d["config"]["host_worker_processes_per_gpu"])
PY
}
run w10a --workers 10
run w12a --workers 12
run w14a --workers 14
run w8a --workers 8
run w10b --workers 10
run w12b --workers 12
run w14b --workers 14
run w7s2 --workers 7 --streams 2

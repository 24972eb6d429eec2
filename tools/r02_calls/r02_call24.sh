#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r02w
mkdir -p $out
W5=$PWD/make_prg_amd/_lib/variants/libmprg_hip_w5.so
run() { tag=$1; shift; "$@" python bench.py --no-cpu-baseline --no-end-to-end --steps 12 --warmup 2 > $out/$tag.json 2> $out/$tag.err; python - $out/$tag.json $tag <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "excl device", d["roofline"]["exclusive_pass"]["device_ms"])
PY
}
run warm env
for rep in 1 2 3; do
run base_$rep env
run w5_$rep env MPRG_HIP_LIB=$W5
done

#!/bin/bash
# round 5, call 30: which of call 28's changes costs config C its 4 %?  The same bench with variant builds (test-only): the group-per-row
# k_cluster_hamming off (CFH_WIDE beyond any view), the gap runs by 2 048-column segments and 8 cells per trip as before
out=gpurun_out/r05_c30; mkdir -p $out
run() {
  label=$1; shift
  env "$@" timeout 500 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg > $out/$label.json 2> $out/$label.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); c=d["config"]
    print("$label", d["value"], d["ms_per_step"], "bad", c["verified"]["mismatches"])
except Exception as e: print("$label failed", e)
PY
}
L=$PWD/make_prg_amd/_lib
for rep in 1 2; do
  run default_$rep X=1
  run cfh_$rep MPRG_HIP_LIB=$L/libmprg_hip_cfh.so
  run gr_$rep MPRG_HIP_LIB=$L/libmprg_hip_gr.so
done
for v in default cfh gr; do
  lib=$L/libmprg_hip.so; [ $v != default ] && lib=$L/libmprg_hip_$v.so
  MPRG_HIP_LIB=$lib MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_$v.txt 2>&1
  echo $v; grep -E "per launch mprg_(partition|cluster_further)|device time|mprg_(partition|cluster_further|ungap_dedupe) " $out/profile_$v.txt | cut -c1-200
done

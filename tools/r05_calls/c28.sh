#!/bin/bash
# round 5, call 28: config D after k_cluster_majority_big, 16-row chunks of wide problems in k_cluster_hamming, 8-row chunks of wide
# views in the row kernels, 512-column gap-run segments, the polled flag in k_partition's pass B; config C's entry points beside it
out=gpurun_out/r05_c28; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt
timeout 600 python tools/config_d_profile.py --passes 4 > $out/config_d_timing.txt 2>&1; tail -22 $out/config_d_timing.txt | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 tools/config_d_profile.py --passes 1 --no-events > $out/run_stats.txt 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/rocprofv3_kernel_stats.csv; head -18 $f | cut -c1-50,150-240
rm -rf $out/prof
MPRG_PROFILE_ALL_LAUNCHES=1 MPRG_BACKEND=runtime timeout 600 python tools/forest_profile.py 7500 2 > $out/profile_7500.txt 2>&1
grep -E "per launch mprg_(partition|ungap|cluster_further)|device time|mprg_" $out/profile_7500.txt | head -16 | cut -c1-200
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-single-worker-leg --no-cli-leg --no-shard-projection --no-deep-leg > $out/bench.json 2> $out/bench.err
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1]); c=d["config"]
print("value", d["value"], d["ms_per_step"], "bad", c["verified"]["mismatches"])
PY

#!/bin/bash
# GPU-box helper: KMeans without scratch memory — parity, exclusive pass per workgroup size, occupancy, 10-worker bench
export TMPDIR=/tmp
out=gpurun_out/r02l
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_b.py tests/test_gpu_ddeep.py -x -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.txt
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
for t in 256 128 512; do
  MPRG_KM_THREADS=$t timeout 600 python bench.py $inproc > $out/threads_$t.json 2> $out/threads_$t.err
done
python - <<'PY'
import json
for t in (256, 128, 512):
    try:
        d = json.loads(open(f"gpurun_out/r02l/threads_{t}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(t, "device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"]) for k in r["kernels"][:6]], "frac", r["frac"], "verified", d["config"]["verified"]["mismatches"])
    except Exception as e:
        print(t, "failed", e)
PY
for set in "OccupancyPercent" "TA_BUSY_avr GRBM_GUI_ACTIVE" "VALUBusy"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$tag -- python3 bench.py $inproc > $out/pmc_$tag.json 2> $out/pmc_$tag.err
  f=$(find $out/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k in ("k_kmeans_restart", "k_kmeans_select", "k_kmeans_prepare_lds", "k_partition", "k_partition_fused", "k_ungap_hash", "k_ungap_dedupe", "k_cluster_majority"):
    for c, (n, s) in agg.get(k, {}).items():
        print(k, c, "launches", n, "sum", s, "mean", s / max(n, 1))
PY
  rm -rf $out/pmc_$tag
done
python bench.py --no-cpu-baseline --no-end-to-end > $out/bench10.json 2> $out/bench10.err; cut -c1-200 $out/bench10.json
python bench.py --no-cpu-baseline --no-end-to-end > $out/bench10b.json 2> $out/bench10b.err; cut -c1-200 $out/bench10b.json

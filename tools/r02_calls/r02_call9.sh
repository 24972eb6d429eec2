#!/bin/bash
# GPU-box helper: GPU tests on the current tree, KMeans workgroup-size sweep, SQ counters of the KMeans kernel
export TMPDIR=/tmp
out=gpurun_out/r02i
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.txt
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
for t in 256 128 64; do
  MPRG_KM_THREADS=$t timeout 600 python bench.py $inproc > $out/threads_$t.json 2> $out/threads_$t.err
done
python - <<'PY'
import json
for t in (256, 128, 64):
    try:
        d = json.loads(open(f"gpurun_out/r02i/threads_{t}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(t, "device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"]) for k in r["kernels"][:6]], "frac", r["frac"])
    except Exception as e:
        print(t, "failed", e)
PY
rocprofv3 -L > $out/counters_avail.txt 2>&1
for set in "VALUBusy" "MemUnitBusy" "MemUnitStalled" "OccupancyPercent" "L2CacheHit" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$tag -- python3 bench.py $inproc > $out/pmc_$tag.json 2> $out/pmc_$tag.err
  f=$(find $out/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k in ("k_kmeans_restart", "k_kmeans_select", "k_ungap_dedupe", "k_partition_fused", "k_cluster_majority", "k_kmeans_prepare_lds"):
    for c, (n, s) in agg.get(k, {}).items():
        print(k, c, "launches", n, "sum", s, "mean", s / max(n, 1))
PY
  rm -rf $out/pmc_$tag
done

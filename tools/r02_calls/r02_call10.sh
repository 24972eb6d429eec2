#!/bin/bash
# GPU-box helper: KMeans workgroup sizes up to 1024 threads, memory-side counters of the KMeans kernel
export TMPDIR=/tmp
out=gpurun_out/r02j
mkdir -p $out
inproc="--workers 0 --streams 1 --batch 3000 --no-cpu-baseline --no-end-to-end --steps 1 --warmup 1"
for t in 256 512 1024; do
  MPRG_KM_THREADS=$t timeout 600 python bench.py $inproc > $out/threads_$t.json 2> $out/threads_$t.err
done
python - <<'PY'
import json
for t in (256, 512, 1024):
    try:
        d = json.loads(open(f"gpurun_out/r02j/threads_{t}.json").read().strip().splitlines()[-1])
        r = d["roofline"]
        print(t, "device_ms", r["exclusive_pass"]["device_ms"], [(k["entry_point"], k["ms"]) for k in r["kernels"][:6]], "frac", r["frac"], "verified", d["config"]["verified"]["mismatches"])
    except Exception as e:
        print(t, "failed", e)
PY
for t in 256 1024; do
for set in "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM" "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" "OccupancyPercent" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '_')_$t
  MPRG_KM_THREADS=$t timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$tag -- python3 bench.py $inproc > $out/pmc_$tag.json 2> $out/pmc_$tag.err
  f=$(find $out/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $t <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k in ("k_kmeans_restart", "k_ungap_hash", "k_ungap_dedupe", "k_partition_fused", "k_cluster_majority"):
    for c, (n, s) in agg.get(k, {}).items():
        print("threads", sys.argv[2], k, c, "launches", n, "sum", s, "mean", s / max(n, 1))
PY
  rm -rf $out/pmc_$tag
done
done

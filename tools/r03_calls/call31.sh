#!/bin/bash
# SQ / memory-side counters of the KMeans fit kernels as they stand (small workgroup form + general form)
export TMPDIR=/tmp
out=gpurun_out/r03_c31
mkdir -p $out
inproc="--workers 0 --streams 1 --batch 8192 --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --steps 1 --warmup 1"
rocprofv3 -L > $out/counters_avail.txt 2>&1
for set in "VALUBusy" "MemUnitBusy" "MemUnitStalled" "OccupancyPercent" "LDSBankConflict" "ALUStalledByLDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "SQ_WAVES SQ_INSTS_FLAT"; do
  tag=$(echo $set | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$tag -- python3 bench.py $inproc > $out/pmc_$tag.json 2> $out/pmc_$tag.err
  f=$(find $out/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $out/counters.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    name = name[5:] if name.startswith("void ") else name
    a = agg[name.split("(")[0]][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    if k.startswith(("k_kmeans_restart_select", "k_partition_fused", "k_cluster_majority", "k_ungap_dedupe")):
        for c, (n, s) in agg[k].items():
            print(k, c, "launches", n, "sum", s, "mean", s / max(n, 1))
PY
  [ -z "$f" ] && echo "$set: no output: $(tail -2 $out/pmc_$tag.err)" | tee -a $out/counters.txt
  rm -rf $out/pmc_$tag
done

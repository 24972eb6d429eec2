#!/bin/bash
# round 6, call 10: SQ / memory-side counters of the LDS form of the KMeans fits (k_kmeans_fit_lds) beside round 5.s profiles/r05/counters
# (one counter set per pass, --kernel-trace --pmc only; bench.py in-process, one stream, 8 192 alignments per pass)
export TMPDIR=/tmp
out=gpurun_out/r06_c10; mkdir -p $out
inproc="--workers 0 --streams 1 --batch 8192 --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 1 --warmup 1"
for set in "VALUBusy" "OccupancyPercent" "LDSBankConflict" "MemUnitBusy" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_ANY" "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" "SQ_WAVES SQ_INSTS_SALU"; do
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
    if k.startswith(("k_kmeans_restart_select", "k_kmeans_fit_lds", "k_kmeans_prepare_lds", "k_cluster_further_one", "k_partition_wave", "k_ungap_hash")):
        for c, (n, s) in agg[k].items():
            print(k, c, "launches", n, "sum", s, "mean", s / max(n, 1))
PY
  [ -z "$f" ] && echo "$set: no output: $(tail -2 $out/pmc_$tag.err)" | tee -a $out/counters.txt
  rm -rf $out/pmc_$tag
done
# host shapes of the 8-GPU shard once more (first pass, 3 750 alignments)
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for shape in "1 2" "2 1" "3 1" "1 3"; do
  set -- $shape
  timeout 600 python bench.py $quick --batch 3750 --workers $1 --streams $2 --first-pass > $out/bench3750_w$1_s$2.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench3750_w$1_s$2.json').read().strip().splitlines()[-1]); print('3750 first pass workers $1 engines $2:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'])"
done

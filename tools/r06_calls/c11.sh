#!/bin/bash
# round 6, call 11: the pairs that need a distance listed PER CENTRE (neighbouring lanes read the same centre row), inertia in cluster order
out=gpurun_out/r06_c11; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kmeans or relocation" > $out/pytest_kmeans.txt 2>&1; tail -1 $out/pytest_kmeans.txt
for n in 7500 3750; do
  timeout 600 python tools/forest_profile.py $n 3 > $out/profile${n}.txt 2>&1
  grep -E "device time|mprg_kmeans|mprg_cluster_loop" $out/profile${n}.txt | grep -v "per launch" | cut -c1-200
done
inproc="--workers 0 --streams 1 --batch 8192 --no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 1 --warmup 1"
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_BUSY_avr GRBM_GUI_ACTIVE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
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
    if k.startswith(("k_kmeans_restart_select", "k_kmeans_fit_lds")):
        for c, (n, s) in agg[k].items():
            print(k, c, "launches", n, "sum", s)
PY
  rm -rf $out/pmc_$tag
done
quick="--no-cpu-baseline --no-end-to-end --no-cli-leg --no-single-worker-leg --no-shard-projection --no-deep-leg --steps 8"
for i in 1 2; do
  timeout 600 python bench.py $quick > $out/bench_$i.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench_$i.json').read().strip().splitlines()[-1]); print('30000:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], d['roofline']['frac'], [(k['entry_point'], k['ms']) for k in d['roofline']['kernels'][:3]])"
done
for cfg in "3750 3" "3750 4" "3750 1" "7500 3" "7500 4" "15000 4"; do
  set -- $cfg
  timeout 600 python bench.py $quick --batch $1 --workers $2 --first-pass > $out/bench$1_w$2.json 2> $out/bench_err.txt
  python -c "import json,sys; d=json.loads(open('$out/bench$1_w$2.json').read().strip().splitlines()[-1]); print('$1 first pass workers $2:', d['value'], d['ms_per_step'], d['config']['verified']['mismatches'], d['config']['host_worker_processes_per_gpu'], d['config']['streams_per_worker'])"
done

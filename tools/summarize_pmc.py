"""Turn rocprofv3 --pmc CSVs (separate FETCH_SIZE and WRITE_SIZE passes) into profiles/<round>/pmc_summary.json.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 following /opt/skills/guides/MI355X_MICROARCH.md §HBM
(FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read).
usage: python tools/summarize_pmc.py <fetch.csv> <write.csv> <bench.json of the same run> <out.json>"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
alg = {k.replace("mprg_", "k_"): v for k, v in bench["config"]["kernels"].items()}
names = {"k_kmeans_restart": "mprg_kmeans_restarts", "k_column_masks": "mprg_column_masks", "k_partition": "mprg_partition",
         "k_ungap_dedupe": "mprg_ungap_dedupe", "k_emit_alleles": "mprg_emit_alleles"}
out = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    calls = max(fetch[k][0], write[k][0], 1)
    f_kb, w_kb = fetch[k][1] / max(fetch[k][0], 1), write[k][1] / max(write[k][0], 1)
    rec = dict(launches=calls, fetch_size_kib_per_launch=round(f_kb, 1), write_size_kib_per_launch=round(w_kb, 1),
               hbm_bytes_per_launch=round((2 * f_kb + w_kb) * 1024))
    b = bench["config"]["kernels"].get(names.get(k, ""))
    if b and b.get("GBps"):
        alg_bytes = b["GBps"] * 1e6 * b["ms"] / b["calls"]
        rec["algorithmic_bytes_per_launch"] = round(alg_bytes)
        rec["traffic_over_algorithmic"] = round(rec["hbm_bytes_per_launch"] / alg_bytes, 3)
    out[k] = rec
json.dump(dict(source=dict(fetch=sys.argv[1], write=sys.argv[2], bench=sys.argv[3]),
               config=dict(batch=bench["config"]["batch_per_gpu"], streams=bench["config"].get("host_threads_streams_per_gpu")),
               kernels=out), open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))

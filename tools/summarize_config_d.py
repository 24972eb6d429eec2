"""profiles/<round>/config_d/: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of tools/config_d_profile.py -> one table.
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, calibrated on k_column_masks);
algorithmic bytes per kernel: the cells (1 byte each) the kernel's step visits, as SURVEY.md §8(d) counts them.
usage: summarize_config_d.py <dir> [rows cols]"""
import collections, csv, gzip, json, os, sys
d = sys.argv[1]
S, C = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10_000, 20_000)
cells = S * C
def kname(raw):
    n = raw.split("(")[0].strip()
    return (n[5:] if n.startswith("void ") else n).split("<")[0]
stats = {}
for r in csv.DictReader(open(os.path.join(d, "rocprofv3_kernel_stats.csv"))):
    n = kname(r["Name"]); a = stats.setdefault(n, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
def pmc(name):
    agg = collections.defaultdict(float)
    with gzip.open(os.path.join(d, name), "rt") as fh:
        for r in csv.DictReader(fh):
            agg[kname(r["Kernel_Name"])] += float(r["Counter_Value"])
    return agg
fetch, write = pmc("pmc_FETCH_SIZE.csv.gz"), pmc("pmc_WRITE_SIZE.csv.gz")
# cells each kernel visits in the config-D build (root view + its one child view, S x C each; the child is the selected view)
ALG = {"k_column_masks": 2 * cells, "k_gap_runs": 2 * cells, "k_partition": 2 * cells, "k_ungap_hash": cells, "k_ungap_hash_u": cells,
       "k_ungap_dedupe": cells, "k_dedupe_scan_big": None, "k_cluster_majority": cells, "k_cluster_majority_big": cells, "k_cluster_hamming": cells, "k_emit_alleles": None, "k_ingest": cells}
# bytes a kernel cannot avoid moving (reads + writes of its step), where that differs from the cells visited
MUST = {"k_ingest": 3 * cells, "k_ungap_hash": 3 * cells, "k_column_masks": 2 * cells, "k_gap_runs": 2 * cells, "k_cluster_majority_big": cells,
        "k_cluster_majority": cells, "k_cluster_hamming": cells, "k_ungap_hash_u": cells}
rows = []
for n, (calls, ns) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if not n.startswith("k_") or ns < 50_000:
        continue
    hbm = (2 * fetch.get(n, 0) + write.get(n, 0)) * 1024
    alg = ALG.get(n)
    rows.append(dict(kernel=n, launches=calls, ms=round(ns / 1e6, 3), hbm_MB=round(hbm / 1e6, 1), hbm_GBps=round(hbm / ns, 1) if ns else None,
                     hbm_frac_of_8TBps=round(hbm / ns / 8000, 4), algorithmic_MB=alg and round(alg / 1e6, 1),
                     algorithmic_GBps=alg and round(alg / ns, 1), traffic_over_algorithmic=alg and round(hbm / alg, 2),
                     must_move_MB=MUST.get(n) and round(MUST[n] / 1e6, 1), traffic_over_must_move=MUST.get(n) and round(hbm / MUST[n], 2)))
json.dump(rows, open(os.path.join(d, "kernels.json"), "w"), indent=1)
print("| kernel | launches | ms | HBM MB (2 x FETCH + WRITE) | HBM GB/s | frac of 8 TB/s | algorithmic MB (cells visited) | traffic / algorithmic | must move MB (reads + writes) | traffic / must move |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r['kernel']}` | {r['launches']} | {r['ms']} | {r['hbm_MB']} | {r['hbm_GBps']} | {r['hbm_frac_of_8TBps']} | {r['algorithmic_MB'] or '—'} | {r['traffic_over_algorithmic'] or '—'} | {r['must_move_MB'] or '—'} | {r['traffic_over_must_move'] or '—'} |")

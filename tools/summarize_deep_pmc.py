"""profiles/r05/deep/: one deep alignment under rocprofv3 — kernel stats + FETCH_SIZE / WRITE_SIZE passes (tools/r05_calls/c13.sh) and
tools/deep_profile.py's JSON of the same sources -> per kernel: time, HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB counters: the
guide's gfx950 correction for wide reads) and, for the wide KMeans fits, the algorithmic bytes 8 D V (iterations + n_init) of the fits
they ran, the achieved algorithmic GB/s and its fraction of the 8 TB/s HBM roofline.
usage: summarize_deep_pmc.py <kernel_stats.csv> <fetch.csv.gz> <write.csv.gz> <deep_profile.json> <out.json>"""
import collections, csv, gzip, json, sys


def kname(raw):
    name = raw.split("(")[0].strip()
    return (name[5:] if name.startswith("void ") else name).split("<")[0]


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    with gzip.open(path, "rt") as fh:
        for r in csv.DictReader(fh):
            a = agg[kname(r["Kernel_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg


stats_csv, fetch_gz, write_gz, deep_json, out_json = sys.argv[1:6]
stats = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(stats_csv)):
    a = stats[kname(r["Name"])]
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
fetch, write = per_kernel(fetch_gz), per_kernel(write_gz)
deep = json.load(open(deep_json))
ep = {e["entry_point"]: e for e in deep["passes"][0].get("entry_points", [])}
total_ns = sum(v[1] for k, v in stats.items() if k.startswith("k_"))
rows = []
for k, (calls, ns) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if not k.startswith("k_") or ns < 0.002 * total_ns:
        continue
    f_kib, w_kib = fetch.get(k, [0, 0.0])[1], write.get(k, [0, 0.0])[1]
    rows.append(dict(kernel=k, launches=calls, ms=round(ns / 1e6, 3), share=round(ns / total_ns, 4),
                     hbm_bytes=round((2 * f_kib + w_kib) * 1024), hbm_GBps=round((2 * f_kib + w_kib) * 1024 / max(ns, 1) , 3)))
wide = ep.get("mprg_kmeans_fit_wide")
roof = None
if wide and wide.get("algorithmic_bytes"):
    WIDE = ("k_kmeans_restart_wide", "k_kmeans_select_list", "k_kmeans_select_only_list", "k_kmeans_predict_list", "k_kmeans_predict_finish")
    ns = sum(stats[k][1] for k in WIDE if k in stats)
    hbm = sum(r["hbm_bytes"] for r in rows if r["kernel"] in WIDE)
    roof = dict(entry_point="mprg_kmeans_fit_wide", kernels="k_kmeans_restart_wide + k_kmeans_select_only_list + k_kmeans_predict_list (+ _finish)", bound="hbm",
                algorithmic_bytes=wide["algorithmic_bytes"], ms_events=wide["ms"], ms_rocprofv3=round(ns / 1e6, 3),
                achieved=round(wide["algorithmic_bytes"] / max(ns, 1), 3), peak=8000.0, unit="GB/s",
                frac=round(wide["algorithmic_bytes"] / max(ns, 1) / 8000.0, 6), traffic=hbm,
                traffic_over_algorithmic=round(hbm / wide["algorithmic_bytes"], 4),
                note="algorithmic = 8 D V (Elkan iterations + n_init) per fit run (SURVEY.md §8d), incl. the fits of rounds the loop "
                     "never reached (every round's fit goes out at once); traffic = 2 x FETCH_SIZE + WRITE_SIZE of the two kernels")
json.dump(dict(config=deep["config"], wall_ms=deep["passes"][-1]["wall_ms"], nodes=deep["passes"][-1]["nodes"], fits=deep["passes"][-1]["fits"],
               prg_sha256=deep["passes"][-1]["prg_sha256"], roofline=roof, kernels=rows), open(out_json, "w"), indent=1)
print(json.dumps(roof))
for r in rows[:10]:
    print(r)

"""Turn rocprofv3 outputs of ONE commit into profiles/<round>/pmc_summary.json:
  kernel stats CSV (--kernel-trace --stats)   -> average duration per kernel
  --pmc FETCH_SIZE and --pmc WRITE_SIZE CSVs  -> HBM-side bytes per launch (separate passes, as the guide prescribes)
  bench JSON of the run under the profiler     -> algorithmic bytes per launch (roofline.kernels, exclusive pass)
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section: x2); narrower access patterns are uncalibrated there, so both the raw
and the doubled figure are kept and the summary says which one the ratio uses (x2 — calibrated here on k_column_masks'
4-byte-per-lane stream: 2 x FETCH_SIZE = 1.00-1.10 x its byte count, profiles/r02/masks_config_d/summary.json).
usage: python tools/summarize_pmc.py <kernel_stats.csv> <fetch.csv> <write.csv> <bench.json of the same run> <out.json>"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import KERNEL_OF, source_digest  # noqa: E402


def kname(raw: str) -> str:
    """'void k_x<36>(long const*, ...)' -> 'k_x' (template instantiations of one kernel are summed)."""
    name = raw.split("(")[0].strip()
    if name.startswith("void "):
        name = name[5:]
    return name.split("<")[0]


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = kname(r["Kernel_Name"])
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


def main():
    stats_csv, fetch_csv, write_csv, bench_json, out_json = sys.argv[1:6]
    stats = {}
    for r in csv.DictReader(open(stats_csv)):          # template instantiations of one kernel: calls and time added up
        n_ = kname(r["Name"])
        if n_ in stats:
            a = stats[n_]
            calls = int(a["Calls"]) + int(r["Calls"])
            tot = float(a["TotalDurationNs"]) + float(r["TotalDurationNs"])
            a.update(Calls=str(calls), TotalDurationNs=str(tot), AverageNs=str(tot / calls), Percentage=str(float(a["Percentage"]) + float(r["Percentage"])))
        else:
            stats[n_] = dict(r)
    fetch, write = per_kernel(fetch_csv), per_kernel(write_csv)
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    alg = {}
    for k in bench["roofline"]["kernels"]:
        for kernel in k["kernel"].replace("(", " ").replace(")", " ").replace("+", " ").replace(",", " ").split():
            if kernel.startswith("k_"):
                alg.setdefault(kernel, k)
    out = {}
    for name in sorted(set(fetch) | set(write) | set(stats)):
        if not name.startswith("k_"):
            continue
        rec = {}
        if name in stats:
            rec.update(launches=int(stats[name]["Calls"]), avg_us=round(float(stats[name]["AverageNs"]) / 1e3, 2),
                       share_of_kernel_time_pct=round(float(stats[name]["Percentage"]), 2))
        f_kb = fetch[name][1] / max(fetch[name][0], 1) if name in fetch else None
        w_kb = write[name][1] / max(write[name][0], 1) if name in write else None
        if f_kb is not None and w_kb is not None:
            rec.update(fetch_size_kib_per_launch=round(f_kb, 1), write_size_kib_per_launch=round(w_kb, 1),
                       hbm_bytes_per_launch_raw=round((f_kb + w_kb) * 1024), hbm_bytes_per_launch=round((2 * f_kb + w_kb) * 1024))
        if name in out:
            continue
        out[name] = rec
    # algorithmic bytes are attributed per ENTRY POINT (several kernels): sum the kernels of one entry point
    groups = collections.defaultdict(list)
    for k in bench["roofline"]["kernels"]:
        label = KERNEL_OF.get(k["entry_point"], k["kernel"])        # the kernels behind an entry point, as bench.py lists them today
        names = [w for w in label.replace("(", " ").replace(")", " ").replace("+", " ").replace(",", " ").split() if w.startswith("k_")]
        groups[k["entry_point"]] = (names, k)
    entry = {}
    # the profiled command runs several passes over the batch (warm-up, timed steps, the exclusive pass); the bench line's
    # per-kernel figures are ONE pass (the exclusive one): compare per pass.  The number of passes comes from the dominant
    # entry point, whose kernel is launched exactly once per call.
    top_names, top = groups[bench["roofline"]["entry_point"]]
    passes = max(1, round(sum(out[n_]["launches"] for n_ in top_names if n_ in out) / max(top["launches"], 1)))
    for ep, (names, k) in groups.items():
        hbm = sum(out.get(n, {}).get("hbm_bytes_per_launch", 0) * out.get(n, {}).get("launches", 0) for n in names) / passes
        hbm_raw = sum(out.get(n, {}).get("hbm_bytes_per_launch_raw", 0) * out.get(n, {}).get("launches", 0) for n in names) / passes
        us = sum(out.get(n, {}).get("avg_us", 0) * out.get(n, {}).get("launches", 0) for n in names) / passes
        rec = dict(kernels=names, passes_profiled=passes, launches_per_pass=k["launches"], bench_ms_per_pass=k["ms"],
                   rocprof_ms_per_pass=round(us / 1e3, 3),
                   algorithmic_bytes_per_pass=k["algorithmic_bytes_per_launch"] and round(k["algorithmic_bytes_per_launch"] * k["launches"]),
                   hbm_bytes_per_pass=round(hbm), hbm_bytes_per_pass_uncorrected=round(hbm_raw))
        if rec["algorithmic_bytes_per_pass"]:
            rec["traffic_over_algorithmic"] = round(hbm / rec["algorithmic_bytes_per_pass"], 3)
            rec["achieved_GBps_rocprof"] = round(rec["algorithmic_bytes_per_pass"] / max(us, 1e-9) * 1e-3, 2)
            rec["frac_of_8TBps"] = round(rec["achieved_GBps_rocprof"] / 8000, 5)
        entry[ep] = rec
    json.dump(dict(source_digest=source_digest(), fetch_correction="2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes)",
                   sources=dict(stats=stats_csv, fetch=fetch_csv, write=write_csv, bench=bench_json),
                   bench_config=dict(batch=bench["config"]["alignments_per_step"], workers=bench["config"]["host_worker_processes_per_gpu"],
                                     streams=bench["config"]["streams_per_worker"]),
                   kernels=out, entry_points=entry), open(out_json, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()

"""From a rocprofv3 --kernel-trace CSV: how many kernels run at the same time, and how much of the wall time has none.
usage: trace_concurrency.py <kernel_trace.csv> [first_step last_step streams]
The window runs from the end of step `first_step` to the end of step `last_step` (steps counted by k_emit_alleles launches, one per
step and stream), e.g. 2 6 for bench.py --warmup 2 --steps 4: the timed steps."""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
first, last, streams = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2, 6, 1)
ev = []
names = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r["Kernel_Name"].split("(")[0].replace("void ", "")))
emits = sorted(e for _, e, n in ev if n.startswith("k_emit_alleles"))
t_lo, t_hi = emits[first * streams - 1], emits[last * streams - 1]
ev = [(max(s, t_lo), min(e, t_hi), n) for s, e, n in ev if e > t_lo and s < t_hi]
points = []
for s, e, _ in ev:
    points.append((s, 1)); points.append((e, -1))
points.sort()
level_time = collections.Counter()
lvl, last = 0, t_lo
for t, d in points:
    level_time[lvl] += t - last
    lvl += d; last = t
wall = t_hi - t_lo
busy = sum(e - s for s, e, _ in ev)
print(f"window {wall / 1e6:.1f} ms, {len(ev)} kernels, sum of kernel durations {busy / 1e6:.1f} ms = {busy / wall:.2f} x wall")
for k in sorted(level_time):
    print(f"  {k} kernels in flight: {100 * level_time[k] / wall:5.1f} % of the window")
by = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    by[n][0] += 1; by[n][1] += e - s
for n, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {n[:60]:60s} {c:6d} launches, mean {d / c / 1e3:8.1f} us, total {d / 1e6:8.1f} ms")

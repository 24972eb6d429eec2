"""Reads a rocprofv3 kernel trace (csv): over the last `frac` of the run, the wall time, the time at least one kernel ran, the sum of
kernel durations, and the same per queue.  usage: trace_overlap.py <kernel_trace.csv> [frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows)
t_lo = ev[0][0] + (1 - frac) * (ev[-1][1] - ev[0][0])
ev = [e for e in ev if e[0] >= t_lo]
wall = ev[-1][1] - ev[0][0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
for s, e, _, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in ev)
print(f"kernels {len(ev)}  wall {wall/1e6:.1f} ms  some kernel running {busy/1e6:.1f} ms ({100*busy/wall:.0f} %)  sum of durations {tot/1e6:.1f} ms  (overlap x{tot/busy:.2f})")
perq = collections.defaultdict(lambda: [0, 0])
for s, e, q, _ in ev:
    perq[q][0] += e - s; perq[q][1] += 1
for q, (d, n) in sorted(perq.items()):
    print(f"  queue {q}: {n} kernels, {d/1e6:.1f} ms")
byname = collections.Counter()
for s, e, q, nme in ev:
    byname[nme[:40]] += e - s
for nme, d in byname.most_common(8):
    print(f"  {d/1e6:8.1f} ms  {nme}")

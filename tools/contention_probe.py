"""GPU-box helper: where does a host worker's step go when W workers share the GPU?  Every worker wraps its backend's
entry points with wall-clock timers (time inside = API overhead + waiting for the device) and runs the bench step.
usage: python tools/contention_probe.py W [loci per worker]   (env such as GPU_MAX_HW_QUEUES is inherited by the workers)"""
import multiprocessing as mp
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(w, W, n, bar, q):
    from bench import make_batch
    msas = make_batch(list(range(w, n * W, W)), 1)[1]
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import ForestEngine
    be = HipBackend(0, own_stream=True)
    acc = {}

    def wrap(name):
        f = getattr(be, name)

        def g(*a, **k):
            t = time.perf_counter()
            r = f(*a, **k)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
            return r
        setattr(be, name, g)
    for nm in ("call", "download", "upload", "empty", "zeros", "synchronize", "grown"):
        wrap(nm)
    eng = ForestEngine(be, max_nesting=5, min_match_length=7)
    with be.on_stream():
        eng.load(msas)
        be.synchronize()
        out = []
        for step in range(3):
            bar.wait()
            acc.clear()
            t0 = time.perf_counter()
            eng.run_forest()
            t1 = time.perf_counter()
            eng.assemble_prgs(as_bytes=True)
            be.synchronize()
            t2 = time.perf_counter()
            out.append(dict(step=t2 - t0, forest=t1 - t0, assemble=t2 - t1, **{k: round(v, 4) for k, v in acc.items()}))
    q.put((w, out[1:]))


if __name__ == "__main__":
    W = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    ctx = mp.get_context("fork")
    bar, q = ctx.Barrier(W), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(w, W, n, bar, q)) for w in range(W)]
    [p.start() for p in ps]
    res = sorted(q.get() for _ in ps)
    [p.join() for p in ps]
    keys = ["step", "forest", "assemble", "call", "download", "upload", "empty", "zeros", "synchronize"]
    print("W =", W, "loci per worker =", n, " GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
    for w, out in res[:3] + res[-1:]:
        for o in out[-1:]:
            print("worker", w, " ".join(f"{k} {1e3 * o.get(k, 0):.0f}" for k in keys), "(ms)")
    last = [out[-1] for _, out in res]
    print("mean over workers:", " ".join(f"{k} {1e3 * sum(o.get(k, 0) for o in last) / len(last):.0f}" for k in keys), "(ms)")
    print("max step", round(1e3 * max(o["step"] for o in last)), "ms ->", round(W * n / max(o["step"] for o in last)), "MSAs/s")

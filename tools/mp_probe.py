"""GPU-box helper: does host-side process parallelism (W worker processes sharing ONE GPU, each with T streams) beat
T threads in one process?  usage: mp_probe.py <batch> <workers> <threads> [steps]"""
import multiprocessing as mp
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(wi, W, batch, T, steps, bar, q):
    from bench import make_batch
    from concurrent.futures import ThreadPoolExecutor
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import ForestEngine
    msas = make_batch(list(range(wi, batch, W)), 1)[1]
    bes = [HipBackend(0, own_stream=True) for _ in range(T)]
    engs = [ForestEngine(b, 5, 7) for b in bes]
    for i, (e, b) in enumerate(zip(engs, bes)):
        with b.on_stream():
            e.load(msas[i::T])
    pool = ThreadPoolExecutor(T)

    def one(i):
        with bes[i].on_stream():
            engs[i].run_forest()
            p = engs[i].assemble_prgs(as_bytes=True)
            bes[i].synchronize()
        return len(p)

    list(pool.map(one, range(T)))
    bar.wait()
    t0 = time.perf_counter()
    for _ in range(steps):
        list(pool.map(one, range(T)))
    dt = time.perf_counter() - t0
    bar.wait()
    q.put(dt)


if __name__ == "__main__":
    batch, W, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    ctx = mp.get_context("fork")
    bar, q = ctx.Barrier(W + 1), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(i, W, batch, T, steps, bar, q)) for i in range(W)]
    for p in ps:
        p.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for p in ps:
        p.join()
    print(f"batch {batch} workers {W} threads {T}: {batch * steps / dt:.0f} MSAs/s  ({1000 * dt / steps:.0f} ms/step)", flush=True)

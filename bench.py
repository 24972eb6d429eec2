#!/usr/bin/env python3
"""bench.py — MSAs/sec of the from_msa hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path (whole recursion forest + PRG string emission) over one batch of synthetic
config-C alignments (the 30k-gene pan-genome shape of BASELINE.json: ~100 seqs x 1-3 kb, generator in
make_prg_amd/utils/synthetic.py) that is already resident in HBM.  Each rank owns `--batch` alignments (weak
scaling: the directory of MSAs shards with no data-path collective) and builds them with `--workers` host worker
processes that share the rank's GPU (the reference's own parallelism is a process pool over alignments; here the
processes overlap the array-at-a-time host control of one sub-batch with the kernels of the others).
Prints ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workers P] [--streams S]
    --workers 0 runs the same loop inside this process (profiler runs: nothing forks)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def _gen(seed):
    from make_prg_amd.msa import load_alignment_text
    from make_prg_amd.utils.synthetic import synth_config_fasta
    return load_alignment_text(synth_config_fasta("C", seed))


def _oracle_one(seed):
    import oracle.from_msa_oracle as orc
    from make_prg_amd.utils.synthetic import synth_config_fasta
    prg, b, root = orc.build_locus_from_text(synth_config_fasta("C", seed), 5, 7)
    return len(prg)


def make_batch(seeds, procs):
    if procs > 1 and len(seeds) >= 64:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            return pool.map(_gen, seeds, chunksize=8)
    return [_gen(s) for s in seeds]


def cpu_baseline(n_sample, procs):
    """The oracle (CPU restatement of the reference path, `port`) on a bounded sample of the same workload."""
    import multiprocessing as mp
    import oracle.from_msa_oracle as orc
    orc.build_kmeans_lib()
    seeds = list(range(1_000_000, 1_000_000 + n_sample))       # same generator/config, disjoint seeds
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_oracle_one, seeds[:procs], chunksize=1)       # untimed: worker start-up, first-touch, imports
        t0 = time.perf_counter()
        pool.map(_oracle_one, seeds, chunksize=1)
        dt = time.perf_counter() - t0
    return n_sample / dt, dt


def _worker(conn, device, seeds, n_streams, profile_mode, gen_procs=1):
    """One host worker process: owns `n_streams` engines (HIP streams, one host thread each) on GPU `device` and a share
    of the rank's alignments.  The reference's own parallelism is a process pool over MSAs (from_msa `-t`); here the
    processes feed one GPU so that the array-at-a-time host control of several sub-batches overlaps."""
    try:
        from concurrent.futures import ThreadPoolExecutor
        from make_prg_amd.backend import HipBackend
        from make_prg_amd.forest import ForestEngine
        msas = make_batch(seeds, gen_procs)             # forks (if at all) before this process touches the GPU
        n_streams = max(1, min(n_streams, len(msas)))
        bes = [HipBackend(device, own_stream=True) for _ in range(n_streams)]
        engs = [ForestEngine(b, max_nesting=5, min_match_length=7) for b in bes]
        t_ing = time.perf_counter()
        for i, (e, b) in enumerate(zip(engs, bes)):      # ingest: encode + upload; inputs are now resident in HBM
            with b.on_stream():
                e.load(msas[i::n_streams])
            b.synchronize()
        t_ing = time.perf_counter() - t_ing
        pool = ThreadPoolExecutor(n_streams)

        def one(i):
            with bes[i].on_stream():
                engs[i].run_forest()                          # recursion forest: kernels + array-at-a-time host control
                prgs = engs[i].assemble_prgs(as_bytes=True)   # PRG text (ASCII) of every locus of the sub-batch
                bes[i].synchronize()
            return sum(p is not None for p in prgs), sum(len(p) for p in prgs if p)

        conn.send(("ready", t_ing))
        while True:
            cmd, arg = conn.recv()
            if cmd == "steps":
                n_ok = chars = 0
                for _ in range(arg):
                    res = list(pool.map(one, range(n_streams)))
                    n_ok, chars = sum(r[0] for r in res), sum(r[1] for r in res)
                conn.send(("done", (n_ok, chars)))
            elif cmd == "reset":
                for b in bes:
                    b.profile = {} if profile_mode != "none" else None
                    b.profile_only = {"mprg_kmeans_restarts"} if profile_mode == "dominant" else None
                for e in engs:
                    for key in e.counters:
                        e.counters[key] = 0 if key != "arena_bytes" else e.counters[key]
                conn.send(("done", None))
            elif cmd == "report":
                prof = {}
                for b in bes:
                    for k_, v_ in b.profile_summary().items():
                        a = prof.setdefault(k_, dict(calls=0, ms=0.0, bytes=0.0))
                        a["calls"] += v_["calls"]; a["ms"] += v_["ms"]; a["bytes"] += v_["bytes"]
                    b.profile = None
                counters = {k_: sum(e.counters.get(k_, 0) for e in engs) for k_ in engs[0].counters}
                counters["levels"] = max(e.counters["levels"] for e in engs)
                conn.send(("done", (prof, counters)))
            else:
                return
    except BaseException as err:        # the parent must hear about it: there is no silent fallback
        import traceback
        conn.send(("error", f"{type(err).__name__}: {err}\n{traceback.format_exc()}"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=30000,
                    help="alignments per GPU per step (default: the whole 30k-gene pan-genome of BASELINE.json, ~15 GB of HBM)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="alignments for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workers", type=int, default=10, help="host worker processes per GPU (each owns a sub-batch)")
    ap.add_argument("--streams", type=int, default=1, help="host threads / HIP streams per worker process")
    ap.add_argument("--profile", choices=("dominant", "all", "none"), default="all",
                    help="HIP-event timing inside the timed region: the dominant kernel's entry point only, every entry "
                         "point (adds event traffic to every launch), or none")
    ap.add_argument("--gen-procs", type=int, default=0, help="processes per worker that generate its alignments (0 = auto)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MPRG_DIST_BACKEND=gloo + MPRG_DEVICE_MODULO=1: several ranks on ONE GPU, to exercise the multi-rank control flow
    # on a single-GPU box (RCCL refuses two ranks on one device); the driver's runs use neither
    dist_backend = os.environ.get("MPRG_DIST_BACKEND", "nccl")
    if os.environ.get("MPRG_DEVICE_MODULO"):
        import torch
        local_rank %= max(torch.cuda.device_count(), 1)          # device_count() does not initialise the GPU
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 "
                     "--master-port P bench.py --gpus N ...")
    ncpu = os.cpu_count() or 1
    cpu = None
    # CPU baseline first, before this process or its workers touch the GPU (fork-safe, and nothing else is running)
    if world == 1 and not args.no_cpu_baseline:
        n = args.cpu_sample or max(ncpu * 6, 48)
        v, dt = cpu_baseline(n, ncpu)
        cpu = dict(value=round(v, 3), unit="MSAs/s", cores=ncpu, kind="port",
                   sample=f"{n} config-C alignments (seeds 1000000..), oracle/ (Python + C KMeans restatement of the "
                          f"reference path), {ncpu} worker processes, one alignment per task, {dt:.1f}s wall")

    # host workers are forked BEFORE this process initialises the GPU (a forked HIP context is unusable)
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    W = max(0, min(args.workers, args.batch))
    if W > 1:          # stay inside the host: at most half the CPUs and a quarter of the free memory (~4 GiB per worker)
        try:
            avail_kib = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1])
        except Exception:
            avail_kib = 64 << 20
        W = max(1, min(W, ncpu // (2 * max(world, 1)), int(avail_kib / (4 << 20) / 4 / max(world, 1))))
    seeds = [rank * 100_000 + i for i in range(args.batch)]
    gen_procs = args.gen_procs or max(1, min(16, ncpu // (max(W, 1) * max(world, 1))))
    conns, procs = [], []
    for w in range(W):
        a, b = ctx.Pipe()
        pr = ctx.Process(target=_worker, args=(b, local_rank, seeds[w::W], args.streams, args.profile, gen_procs))
        pr.start()
        conns.append(a); procs.append(pr)
    if W == 0:          # --workers 0: the same worker loop on a thread of this process (rocprofv3 runs: nothing forks)
        import threading
        a, b = ctx.Pipe()
        th = threading.Thread(target=_worker, args=(b, local_rank, seeds, args.streams, args.profile), daemon=True)
        th.start()
        conns.append(a)

    def gather(expect="done"):
        out = []
        for i, c in enumerate(conns):
            while not c.poll(5.0):          # a worker that died without a word (killed) must not hang the run
                if procs and not procs[i].is_alive():
                    sys.stderr.write(f"bench worker {i} exited with code {procs[i].exitcode}\n")
                    for pr in procs:
                        pr.terminate()
                    os._exit(1)
            tag, val = c.recv()
            if tag == "error":
                sys.stderr.write(val)
                for pr in procs:
                    pr.terminate()
                os._exit(1)
            assert tag == expect, (tag, expect)
            out.append(val)
        return out

    def command(cmd, arg=None):
        for c in conns:
            c.send((cmd, arg))
        return gather()

    t_ing = max(gather("ready"))

    import torch
    import torch.distributed as dist
    if world > 1:
        torch.cuda.set_device(local_rank)
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend)
    device = torch.device("cuda", local_rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)     # the workers synchronise their own streams before they answer "done"

    command("steps", args.warmup)
    command("reset")
    barrier()
    t0 = time.perf_counter()
    res = command("steps", args.steps)       # every worker runs its K steps back to back; no collective on the data path
    barrier()
    dt = time.perf_counter() - t0
    n_ok = sum(r[0] for r in res)
    reports = command("report")
    for c in conns:
        c.send(("quit", None))
    prof = {}
    for pr_, _ in reports:
        for k_, v_ in pr_.items():
            a = prof.setdefault(k_, dict(calls=0, ms=0.0, bytes=0.0))
            a["calls"] += v_["calls"]; a["ms"] += v_["ms"]; a["bytes"] += v_["bytes"]
    counters = {k_: sum(c_[k_] for _, c_ in reports) for k_ in reports[0][1]}
    counters["levels"] = max(c_["levels"] for _, c_ in reports)
    n_streams = args.streams

    t = torch.tensor([dt], dtype=torch.float64, device=device if dist_backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    total_msas = args.batch * world * args.steps
    value = total_msas / dt_max

    if rank == 0:
        dev_ms = sum(v["ms"] for v in prof.values())
        if not prof:
            prof = {"mprg_kmeans_restarts": dict(calls=0, ms=0.0, bytes=0.0)}
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        name, d = dom
        launches = d["calls"]
        achieved = (d["bytes"] / max(d["ms"], 1e-9)) * 1e-6          # bytes/ms -> GB/s
        # HBM traffic of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
        # gfx950 correction 2*FETCH+WRITE; tools/summarize_pmc.py): counters cannot be read inside this process, so the
        # committed ratio traffic/algorithmic of the profiled run is applied to this run's algorithmic bytes per launch
        traffic = None
        try:
            import glob
            pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json")))[-1]))
            ratio = pm["kernels"][{"mprg_kmeans_restarts": "k_kmeans_restart"}.get(name, name.replace("mprg_", "k_"))]["traffic_over_algorithmic"]
            traffic = round(ratio * d["bytes"] / max(launches, 1), 1)
        except Exception:
            pass
        roof = dict(bound="hbm", kernel=name, achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 6), traffic=traffic,
                    avg_launch_ms=round(d["ms"] / max(launches, 1), 4), launches=launches,
                    algorithmic_bytes_per_launch=round(d["bytes"] / max(launches, 1), 1))
        kernels = {k: dict(ms=round(v["ms"], 3), calls=v["calls"],
                           GBps=round((v["bytes"] / max(v["ms"], 1e-9)) * 1e-6, 3) if v["bytes"] else None)
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        out = {
            "metric": "MSAs/sec (from_msa, whole node) on 30k-gene pan-genome",
            "value": round(value, 3), "unit": "MSAs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt_max / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 (+f64 KMeans)", "data": "synthetic",
            "config": {"workload": "C: 30k-gene pan-genome shape (S~N(100,20) in [20,300] rows x 1000-3000 cols, "
                                   "SURVEY.md §8d generator), -N 5 -L 7; one step = one resident batch per GPU",
                       "batch_per_gpu": args.batch, "parallelism": f"shard{world}", "host_worker_processes_per_gpu": W, "streams_per_worker": n_streams,
                       "event_timing": args.profile,
                       "step_includes": "recursion forest on device + host control + PRG string emission",
                       "ingest_s_excluded": round(t_ing, 3), "device_ms_per_step": round(dev_ms / args.steps, 3),
                       "levels": counters["levels"] / args.steps, "launches_per_step": counters["launches"] / args.steps,
                       "kmeans_fits_per_step": counters["fits"] / args.steps,
                       "B_alg_bytes_per_step": (counters["cells_all"] + counters["cells_clustered"]
                                                + counters["kmeans_bytes"]) / args.steps,
                       "kernels": kernels},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    for pr in procs:
        pr.join(timeout=30)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

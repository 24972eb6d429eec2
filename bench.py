#!/usr/bin/env python3
"""bench.py — MSAs/sec of the from_msa hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path (whole recursion forest + PRG string emission) over the synthetic config-C set (the
30k-gene pan-genome shape of BASELINE.json: ~100 seqs x 1-3 kb, generator in make_prg_amd/utils/synthetic.py, seeds
0..batch-1) that is already resident in HBM.  With N ranks the SAME `--batch` alignments are sharded over the ranks by
size (longest-processing-time greedy on rows x columns, the rule the CLI uses on file sizes): strong scaling, BASELINE
config 3, no data-path collective (`--weak`: every rank builds all of them).  A rank builds its shard with `--workers` host
worker processes that share the rank's GPU; every table of the recursion lives on the device (make_prg_amd/forest.py), so ONE
worker keeps the device busy — more workers only fill the gaps at the host's waits (`single_worker` in the line: the same run
with one worker).

What the one JSON line holds (rank 0):
  value            K timed steps of all workers, HIP-event timing OFF, barrier + synchronize on both sides, MAX over ranks
  verified         every PRG + node count of the last timed step against tests/golden/config_c_digests.bin (the oracle's
                   answers for seeds 0..29999), outside the timed region; a mismatch makes the run exit non-zero
  roofline         from an untimed EXCLUSIVE pass: worker 0 alone on the device, one stream, its own sub-batch, HIP events
                   around every entry point (so the kernels' times add up to less than that pass's wall time); dominant
                   kernel + the top kernels with their fraction of the HBM roofline
  end_to_end       (N=1) FASTA bytes in memory -> parsed (native batch parser), uploaded, built, PRG + .bin + .gfa bytes in memory
                   (native batch encoders), all workers in parallel: the command line's stages without its files; the CPU
                   baseline's end_to_end_value covers the identical region
  cpu_baseline     (N=1) the oracle (CPU restatement of the reference path, `port`) on a bounded sample, all host cores

Defaults: K = 10 timed steps after W = 2 warm-up steps.  The workers run their steps back to back without a barrier in
between; they leave the start barrier in lockstep (every worker in the same phase: the GPU idles while all of them
assemble PRG strings) and fall out of phase over the first steps, so few steps measure the transient, not the rate
(3 steps: ~745 ms per step, 12 steps: ~670 ms on the same box, profiles/r02/README.md).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workers P] [--streams S] [--weak]
    --workers 0 runs the same loop inside this process (profiler runs: nothing forks)
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
DIGESTS = os.path.join(ROOT, "tests", "golden", "config_c_digests.bin")
KERNEL_OF = {"mprg_kmeans_restarts": "k_kmeans_restart", "mprg_kmeans_fit": "k_kmeans_restart_select (persistent form: k_kmeans_fit)",
             "mprg_kmeans_fit_small": "k_kmeans_restart_select_small", "mprg_kmeans_fit_split": "k_kmeans_restart_one + k_kmeans_select_list", "mprg_kmeans_fit_wave": "k_kmeans_fit_wave", "mprg_kmeans_fit_lds": "k_kmeans_fit_lds",
             "mprg_column_masks": "k_column_masks", "mprg_partition": "k_partition (+ k_partition_wave, k_partition_fused, k_gap_runs, k_pack_scan, k_pack_copy)",
             "mprg_ungap_dedupe": "k_ungap_dedupe (+ k_dedupe_wave, k_ungap_hash, k_ungap_hash_u, k_dedupe_scan_big)", "mprg_emit_alleles": "k_emit_alleles",
             "mprg_cluster_loop[general]": "k_cluster_loop", "mprg_cluster_loop[small]": "k_cluster_loop_small",
             "mprg_cluster_further": "k_cluster_further_one (+ k_cluster_majority, k_cluster_majority_big, k_cluster_hamming)",
             "mprg_kmeans_prepare": "k_kmeans_prepare_lds (+ k_kmeans_prepare, k_kmeans_prepare_tables_tiled, k_kmeans_prepare_tables)"}


def _text(seed):
    from make_prg_amd.utils.synthetic import synth_config_fasta
    return synth_config_fasta("C", seed)


def _gen(seed):
    from make_prg_amd.msa import load_alignment_text
    text = _text(seed)
    return text, load_alignment_text(text, defer_n=True)


def _oracle_one(seed):
    """The CPU leg of one alignment: (seconds for parse + build, seconds for the .bin and .gfa encoders on top)."""
    import oracle.from_msa_oracle as orc
    text = _text(seed)
    t0 = time.perf_counter()
    prg, b, root = orc.build_locus_from_text(text, 5, 7)
    t1 = time.perf_counter()
    orc.encode_prg_bytes(prg)
    orc.gfa_text(prg)
    return t1 - t0, time.perf_counter() - t1


def lpt_parts(seeds, n_parts):
    """Size-balanced split of the alignments `seeds` into n_parts lists: longest-processing-time greedy on rows x columns
    (make_prg_amd.subcommands.from_msa.balanced_parts does the same on file sizes); deterministic."""
    import heapq
    from make_prg_amd.utils.synthetic import config_shape
    cost = []
    for sd in seeds:
        S, C, _ = config_shape("C", sd)
        cost.append((-S * C, sd))
    cost.sort()
    heap = [(0, p) for p in range(n_parts)]
    parts = [[] for _ in range(n_parts)]
    for neg, sd in cost:
        load, p = heapq.heappop(heap)
        parts[p].append(sd)
        heapq.heappush(heap, (load - neg, p))
    return [sorted(p) for p in parts]


def make_batch(seeds, procs):
    if procs > 1 and len(seeds) >= 64:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            pairs = pool.map(_gen, seeds, chunksize=8)
    else:
        pairs = [_gen(s) for s in seeds]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def cpu_baseline(n_sample, procs):
    """The oracle (CPU restatement of the reference path, `port`) on a bounded sample of the same workload."""
    import multiprocessing as mp
    import oracle.from_msa_oracle as orc
    orc.build_kmeans_lib()
    seeds = list(range(1_000_000, 1_000_000 + n_sample))       # same generator/config, disjoint seeds
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_oracle_one, seeds[:procs], chunksize=1)       # untimed: worker start-up, first-touch, imports
        t0 = time.perf_counter()
        parts = pool.map(_oracle_one, seeds, chunksize=1)
        dt = time.perf_counter() - t0
    build = sum(p[0] for p in parts)
    enc = sum(p[1] for p in parts)
    # the pool's wall time covers build + encoders; the build-only rate removes the encoders' share of the CPU seconds
    return n_sample / dt, n_sample / (dt * build / (build + enc)), dt


def source_digest():
    """sha256 over the kernel sources + the host that drives them: ties a committed PMC summary to the code it measured."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "make_prg_amd", "csrc")
    names = [n for n in sorted(os.listdir(d)) if n.endswith((".inc", ".hip", ".h", ".cpp"))]
    for name in names + ["../forest.py", "../engine.py"]:
        with open(os.path.join(d, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _worker(conn, device, seeds, n_streams, gen_procs=1, cli_dir=None, backend_kind=None, first_pass=False):
    """One host worker process: owns `n_streams` engines (HIP streams, one host thread each) on GPU `device` and a share
    of the rank's alignments.  The reference's own parallelism is a process pool over MSAs (from_msa `-t`); here the
    processes feed one GPU so that the array-at-a-time host control of several sub-batches overlaps."""
    try:
        import numpy as np
        from make_prg_amd.backend import make_backend
        from make_prg_amd.forest import ForestEngine
        texts, msas = make_batch(seeds, gen_procs)      # forks (if at all) before this process touches the GPU
        texts_b = [t.encode() for t in texts]           # the end-to-end leg starts from the bytes a file would hold
        e2e_threads = max(1, gen_procs)                 # threads of the native batch stages in that leg (CPUs / workers)
        if cli_dir:                                     # the command-line leg reads the same alignments as FASTA files
            for sd, t in zip(seeds, texts):
                with open(os.path.join(cli_dir, f"gene{sd:05d}.fa"), "w") as fh:
                    fh.write(t)
        n_streams = max(1, min(n_streams, len(msas)))
        # buffers, streams and events from the library's own mprg_rt_* plumbing (MPRG_BACKEND=torch: from torch.cuda); same kernels
        bes = [make_backend(backend_kind, device, own_stream=True) for _ in range(n_streams)]
        engs = [ForestEngine(b, max_nesting=5, min_match_length=7) for b in bes]
        t_ing = time.perf_counter()
        for i, (e, b) in enumerate(zip(engs, bes)):      # ingest: encode + upload; inputs are now resident in HBM
            with b.on_stream():
                e.load(msas[i::n_streams])
            b.synchronize()
        t_ing = time.perf_counter() - t_ing
        last = [None] * n_streams
        # --first-pass: every pass of the timed region builds its sub-batch as a batch seen for the FIRST time — what a rank of a
        # multi-GPU run or a chunk of the command line is: no totals of its own to size buffers from.  The capacities of its levels
        # are predicted from ANOTHER batch (forest.ForestEngine._caps_predicted): a calibration batch of disjoint seeds, built once,
        # untimed (the chunk before, in a real run).
        calib = None
        if first_pass:
            _, cal_msas = make_batch([2_000_000 + q for q in range(max(64, min(1024, len(msas) // max(n_streams, 1))))], gen_procs)
            with bes[0].on_stream():
                ce = ForestEngine(bes[0], max_nesting=5, min_match_length=7)
                ce.load(cal_msas)
                ce.run_forest()
                calib = ce.plan_export()
            bes[0].synchronize()
            del ce, cal_msas

        def enqueue(i):
            if first_pass:
                engs[i]._plan, engs[i].plan_donor = None, calib
            engs[i].forest_enqueue()

        def collect_one(i, fin):
            prgs = fin()
            last[i] = prgs
            return sum(p is not None for p in prgs), sum(len(p) for p in prgs if p is not None)

        def run_steps(n_steps):
            """n_steps passes over every sub-batch, ONE host thread.  An engine's step: wait for its forest (enqueued earlier without a
            wait: forest.forest_enqueue, levels sized from the previous pass's totals, counts on the device), lay the PRG text out and
            start its copy to pinned memory, enqueue the NEXT forest, then collect the previous step's text.  The engines take turns, so
            while the host waits for one, the others' forests are queued on their own streams and fill the device (the first pass of a
            batch has no plan yet: the per-step host runs it inside forest_enqueue)."""
            pending = [None] * n_streams
            totals = [(0, 0)] * n_streams
            if n_steps <= 0:
                return 0, 0
            for i in range(n_streams):
                with bes[i].on_stream():
                    enqueue(i)
            for step in range(n_steps):
                for i in range(n_streams):
                    with bes[i].on_stream():
                        engs[i].forest_finish()
                        # PRG text (ASCII) of every locus of the sub-batch: laid out and written on the device, copied to a pinned
                        # host buffer on the copy stream; collected one step later, so the copy overlaps the next step's kernels
                        fin = engs[i].assemble_prgs(as_bytes=True, lazy=True)
                        if step + 1 < n_steps:
                            enqueue(i)
                    if pending[i] is not None:
                        collect_one(i, pending[i])
                    pending[i] = fin
            for i in range(n_streams):
                totals[i] = collect_one(i, pending[i])
            return sum(t[0] for t in totals), sum(t[1] for t in totals)

        def end_to_end():
            """FASTA text (bytes in memory) -> PRG, .bin and .gfa bytes in memory, nothing resident beforehand: the stages of the
            command line's pipeline (make_prg_amd/pipeline.py) without its files — libmprg's batch parser into a pinned arena,
            one upload, the forest and the PRG text on the device, every locus encoded once by the batch encoders; the natives
            run `e2e_threads` threads per worker."""
            import ctypes
            from make_prg_amd.utils import native
            lib = native.library()
            n = len(texts_b)
            t0 = time.perf_counter()
            ptrs = (ctypes.c_char_p * n)(*texts_b)
            lens = np.fromiter((len(t) for t in texts_b), np.int64, n)
            h = lib.mprg_ingest_open_mem_host(ptrs, lens.ctypes.data, n, e2e_threads)
            info = np.zeros((n, 5), np.int64)
            lib.mprg_ingest_info_host(h, info.ctypes.data)
            status, rows, cols, tbytes, flags = (info[:, k] for k in range(5))
            assert not status.any() and not (flags & 1).any(), "synthetic alignments are plain FASTA with distinct ids"
            sizes = rows * cols
            raw_off, t_off = np.cumsum(sizes) - sizes, np.cumsum(tbytes) - tbytes
            be = bes[0]
            arena_buf, arena = be.pinned(int(sizes.sum()), "e2e")
            titles = np.empty(max(int(tbytes.sum()), 1), np.uint8)
            lib.mprg_ingest_fill_host(h, arena.ctypes.data, raw_off.ctypes.data, titles.ctypes.data, t_off.ctypes.data, e2e_threads)
            t1 = time.perf_counter()
            with be.on_stream():
                eng = ForestEngine(be, max_nesting=5, min_match_length=7)
                eng.load_raw(arena_buf, arena, raw_off, rows, cols, has_n=(flags & 2) != 0)
                t2 = time.perf_counter()
                eng.run_forest()
                fin = eng.assemble_prgs(as_bytes=True, lazy=True)
                fin()
            t3 = time.perf_counter()
            length, base = np.ascontiguousarray(fin.length, np.int64), np.ascontiguousarray(fin.base, np.int64)
            whole = np.frombuffer(fin.buffer, np.uint8)
            ba, bw, ga, gb = (np.zeros(n, np.int64) for _ in range(4))
            crc = np.zeros((n, 3), np.uint32)
            pool_ = lib.mprg_encode_pool_new_host()
            rc = lib.mprg_encode_batch_host(pool_, whole.ctypes.data, base.ctypes.data, length.ctypes.data, n, e2e_threads, 1, 1,
                                            ba.ctypes.data, bw.ctypes.data, ga.ctypes.data, gb.ctypes.data, crc.ctypes.data)
            assert rc == 0 and (bw[length >= 0] >= 0).all() and (gb[length >= 0] >= 0).all()
            n_bytes = int(4 * bw[bw > 0].sum() + gb[gb > 0].sum())
            t4 = time.perf_counter()
            lib.mprg_encode_pool_free_host(pool_)
            lib.mprg_ingest_close_host(h)
            return dict(n=int((length >= 0).sum()), parse_s=t1 - t0, encode_upload_s=t2 - t1, build_s=t3 - t2,
                        encoders_s=t4 - t3, out_bytes=n_bytes)

        def verify():
            if not os.path.exists(DIGESTS):
                return None
            from tests.config_c_full import load_digests, mismatches
            blob = load_digests(DIGESTS)
            bad = []
            for i, e in enumerate(engs):
                sub = seeds[i::n_streams]
                if not sub or max(sub) * 12 + 12 > len(blob) or last[i] is None:
                    return None
                n_nodes = np.bincount(e.tab["msa"], minlength=len(sub))
                bad += mismatches(blob, sub, [None if p is None else bytes(p) for p in last[i]], n_nodes)
            return bad

        # the resident batch is ~10^6 long-lived Python containers (ids, descriptions): keep the cyclic collector from walking
        # them every time a step's short-lived lists trigger a full collection (measured: 130-210 ms pauses every third step)
        import gc
        gc.collect()
        gc.freeze()
        conn.send(("ready", t_ing))
        while True:
            cmd, arg = conn.recv()
            if cmd == "steps":
                t0 = time.perf_counter()
                n_ok, chars = run_steps(arg)                  # every step's text is on the host when the worker answers
                for b in bes:
                    b.synchronize()
                conn.send(("done", (n_ok, chars, time.perf_counter() - t0)))
            elif cmd == "reset":                              # arg: HIP-event timing of every entry point on / off
                for b in bes:
                    b.profile = {} if arg else None
                    b.profile_only = None
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
                conn.send(("done", (prof, counters, len(seeds))))
            elif cmd == "verify":
                conn.send(("done", verify()))
            elif cmd == "e2e":
                conn.send(("done", end_to_end()))
            else:
                return
    except BaseException as err:        # the parent must hear about it: there is no silent fallback
        import traceback
        conn.send(("error", f"{type(err).__name__}: {err}\n{traceback.format_exc()}"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=30000,
                    help="alignments of the job per step (default: the whole 30k-gene pan-genome of BASELINE.json, ~15 GB of HBM); "
                         "sharded over the ranks unless --weak")
    ap.add_argument("--weak", action="store_true", help="every rank builds all --batch alignments (weak scaling)")
    ap.add_argument("--first-pass", action="store_true",
                    help="every timed pass builds its alignments as a batch seen for the first time: level capacities predicted from a "
                         "DIFFERENT (calibration) batch instead of taken from the previous pass of the same batch")
    ap.add_argument("--no-single-worker-leg", action="store_true")
    ap.add_argument("--no-shard-projection", action="store_true", help="skip the runs at the shard sizes of 2 / 4 / 8 GPUs (N=1 only)")
    ap.add_argument("--no-cli-leg", action="store_true", help="skip the file -> file run of the command line (N=1 only)")
    ap.add_argument("--cli-threads", type=int, default=0, help="-t of the command-line leg (0 = the CPUs this process may use, at most 16)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="alignments for the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-deep-leg", action="store_true", help="skip the one-deep-alignment leg (N=1 only)")
    ap.add_argument("--workers", type=int, default=4, help="host worker processes per GPU (each owns a sub-batch); capped by the "
                                                           "CPUs this rank may use")
    ap.add_argument("--streams", type=int, default=0, help="engines (sub-batches on HIP streams of their own, fed by ONE host thread) per worker "
                                                           "process; 0 = by the shard's size (see below)")
    ap.add_argument("--profile-timed", action="store_true",
                    help="HIP events around every entry point INSIDE the timed region too (diagnostic; adds event traffic)")
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
    from make_prg_amd.utils.misc import effective_cpus
    ncpu = effective_cpus()          # affinity and cgroup quota, not the machine's core count
    cpu = None
    # CPU baseline first, before this process or its workers touch the GPU (fork-safe, and nothing else is running)
    if world == 1 and not args.no_cpu_baseline:
        n = args.cpu_sample or max(ncpu * 6, 1000)          # ~9 s on 16 cores: beyond start-up noise (a fixed 1 000-alignment subsample)
        v_e2e, v_build, dt = cpu_baseline(n, ncpu)
        calib = ""
        try:
            c = json.load(open(os.path.join(ROOT, "profiles", "r02", "cpu_calibration.json")))["sets"]["C-sub"]
            calib = (f"; calibration in the build container (8 vCPU, 500 config-C alignments): the REAL reference is "
                     f"{c['reference_over_port']}x the port's speed ({c['reference_msas_per_s']} vs {c['port_msas_per_s']} MSAs/s)")
        except Exception:
            pass
        cpu = dict(value=round(v_build, 3), unit="MSAs/s", cores=ncpu, kind="port", end_to_end_value=round(v_e2e, 3),
                   sample=f"{n} config-C alignments (seeds 1000000..), oracle/ (Python + C KMeans restatement of the "
                          f"reference path), {ncpu} worker processes, one alignment per task, {dt:.1f}s wall; value = FASTA "
                          f"text -> PRG string, end_to_end_value = + .bin and .gfa encoders{calib}")

    # host workers are forked BEFORE this process initialises the GPU (a forked HIP context is unusable)
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    all_seeds = list(range(args.batch))
    # strong scaling (default): the job's alignments are sharded over the ranks by size; every rank verifies its own seeds
    seeds = all_seeds if (args.weak or world == 1) else lpt_parts(all_seeds, world)[rank]
    W = max(0, min(args.workers, len(seeds)))
    if W > 1:          # stay inside the host: the CPUs this process may use and a quarter of the free memory (~4 GiB per worker)
        try:
            avail_kib = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1])
        except Exception:
            avail_kib = 64 << 20
        W = max(1, min(W, max(1, (ncpu - 1) // max(world, 1)), int(avail_kib / (4 << 20) / 4 / max(world, 1))))
    # The shape of a rank's host side follows its shard (measured on MI355X, profiles/r04/host_shapes.md, re-measured in round 5: profiles/r05/host_shapes.txt):
    #   a small shard (what a rank of an 8-GPU run sees: 3 750 alignments per step) is built best by ONE worker process whose single
    #   host thread feeds TWO engines (sub-batches on streams of their own; forests enqueued without waits from the previous pass's
    #   totals, forest.forest_enqueue) with the clustering loop's general and small forms side by side on a side stream:
    #   76.1 k alignments/s against 67.9 k with one engine and 75.2 k with four worker processes;
    #   a big shard by several worker processes with one engine each (30 000: 95.3 k with four; two engines each: 90.5 k).
    # Round 6 (profiles/r06/NOTES.md: host shapes): with the LDS form of the KMeans fits several worker processes of one engine each build
    # such a shard fastest too — first pass, 3 750 alignments: 93.2 k alignments/s with four, 92.1 k with three, 88.4 k with one worker
    # and two engines (round 5's shape: 76.1 k then) — so a small shard keeps the default shape (MPRG_SHARD_WORKERS: for experiments).
    if args.workers == 4 and len(seeds) <= 5000 and W >= 1:          # (an explicit --workers other than the default is respected)
        W = min(W, int(os.environ.get("MPRG_SHARD_WORKERS", "4")))
    if args.streams == 0:
        args.streams = 2 if (W == 1 and len(seeds) <= 12000) else 1
    if W == 1:          # (not --workers 0: the rocprofv3 runs want one kernel at a time)
        os.environ.setdefault("MPRG_KM_SIDE_STREAMS", "1")
    gen_procs = args.gen_procs or max(1, min(16, ncpu // (max(W, 1) * max(world, 1))))
    parts = lpt_parts(seeds, W) if W > 1 else [seeds]
    cli_dir = None
    if world == 1 and not args.no_cli_leg and W >= 1:
        import tempfile
        cli_root = tempfile.mkdtemp(prefix="mprg_bench_cli_")
        cli_dir = os.path.join(cli_root, "msas")
        os.mkdir(cli_dir)
    conns, procs, th = [], [], None
    for w in range(W):
        a, b = ctx.Pipe()
        pr = ctx.Process(target=_worker, args=(b, local_rank, parts[w], args.streams, gen_procs, cli_dir, None, args.first_pass))
        pr.start()
        conns.append(a); procs.append(pr)
    if W == 0:          # --workers 0: the same worker loop on a thread of this process (rocprofv3 runs: nothing forks)
        import threading
        a, b = ctx.Pipe()
        # (this process also runs torch: its worker thread takes its buffers and streams from torch too — one HIP runtime per process)
        th = threading.Thread(target=_worker, args=(b, local_rank, seeds, args.streams, 1, None, "torch", args.first_pass), daemon=True)
        th.start()
        conns.append(a)

    def die(msg):
        sys.stderr.write(msg)
        for pr in procs:
            pr.terminate()
        os._exit(1)

    def gather(which, expect="done"):
        out = []
        for i in which:
            c = conns[i]
            while not c.poll(5.0):          # a worker that died without a word (killed) must not hang the run
                if procs and not procs[i].is_alive():
                    die(f"bench worker {i} exited with code {procs[i].exitcode}\n")
            tag, val = c.recv()
            if tag == "error":
                die(val)
            assert tag == expect, (tag, expect)
            out.append(val)
        return out

    everyone = list(range(len(conns)))

    def command(cmd, arg=None, which=None):
        which = everyone if which is None else which
        for i in which:
            conns[i].send((cmd, arg))
        return gather(which)

    t_ing = max(gather(everyone, "ready"))

    import torch
    import torch.distributed as dist
    # (MPRG_DIST_FORCE=1 under a launcher: the process group — RCCL by default — also for one rank, so that a one-GPU box exercises
    #  the barrier and the reductions of the multi-GPU line)
    grouped = world > 1 or (os.environ.get("MPRG_DIST_FORCE", "0") != "0" and "MASTER_ADDR" in os.environ)
    if grouped:
        torch.cuda.set_device(local_rank)
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend)
    device = torch.device("cuda", local_rank)

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize(device)     # the workers synchronise their own streams before they answer "done"

    command("steps", args.warmup)
    command("reset", args.profile_timed)
    barrier()
    t0 = time.perf_counter()
    res = command("steps", args.steps)       # every worker runs its K steps back to back; no collective on the data path
    barrier()
    dt = time.perf_counter() - t0
    n_ok = sum(r[0] for r in res)
    reports = command("report")
    counters = {k_: sum(c_[k_] for _, c_, _ in reports) for k_ in reports[0][1]}
    counters["levels"] = max(c_["levels"] for _, c_, _ in reports)

    t = torch.tensor([dt], dtype=torch.float64, device=device if (dist_backend == "nccl" and grouped) else "cpu")
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    n_local = torch.tensor([len(seeds)], dtype=torch.float64, device=device if (dist_backend == "nccl" and grouped) else "cpu")
    if grouped:
        dist.all_reduce(n_local, op=dist.ReduceOp.SUM)
    msas_per_step = int(n_local.item())          # strong: --batch; weak: --batch x ranks
    total_msas = msas_per_step * args.steps
    value = total_msas / dt_max

    # ---- outside the timed region: check what was timed, byte for byte
    bad_lists = command("verify")
    verified = None
    if all(b is not None for b in bad_lists):
        bad = sorted(x for b in bad_lists for x in b)
        verified = dict(loci=len(seeds), mismatches=len(bad), first_bad=bad[:10],
                        against="tests/golden/config_c_digests.bin (oracle: sha256(PRG)[:8] + node count per seed)")
        try:          # which reference the digests are tied to, and how much of this data the reference itself decides stably
            tie = json.load(open(os.path.join(ROOT, "tests", "golden", "config_c_reference_tie.json")))
            st = tie["stability"]
            verified["reference_tie"] = dict(
                parity="Haswell-pinned: the REAL reference (make_prg v0.5.0, scikit-learn n_init=10, OMP_NUM_THREADS=1, OPENBLAS_CORETYPE=Haswell) "
                       f"gives these digests for seeds {tie['config_c']['seeds']} ({tie['config_c']['equal_to_digest_fixture']} equal, container run "
                       "oracle/tools/tie_reference.py)",
                stable_loci=st["stable"], unstable_loci=st["unstable"], of=st["stable"] + st["unstable"], families=st["families"],
                note="stable = PRG byte-identical under all five OpenBLAS kernel families; on the rest the real reference's own answer "
                     "depends on the CPU it runs on (exact k-means++ / centre ties decided by the last bits of BLAS sums) — parity is to the pinned run")
        except Exception:
            pass
    if grouped:
        flag = torch.tensor([0 if verified is None else verified["mismatches"]], dtype=torch.float64,
                            device=device if (dist_backend == "nccl" and grouped) else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        if verified is not None:
            verified["mismatches_all_ranks"] = int(flag.item())

    excl = e2e = None
    if rank == 0:
        # ---- exclusive pass: worker 0 alone on the device, HIP events around every entry point
        command("reset", True, which=[0])
        (x_ok, _, x_wall), = command("steps", 1, which=[0])
        (x_prof, x_cnt, x_n), = command("report", which=[0])
        excl = dict(prof=x_prof, counters=x_cnt, wall_ms=1000.0 * x_wall, loci=x_n)
    if world == 1 and not args.no_end_to_end:
        barrier()
        t0 = time.perf_counter()
        parts = command("e2e")
        e2e_s = time.perf_counter() - t0
        e2e = dict(value=round(sum(p["n"] for p in parts) / e2e_s, 3), unit="MSAs/s", seconds=round(e2e_s, 3),
                   region="FASTA bytes in memory -> libmprg's batch parser into a pinned arena (parse_s) -> one upload + device ingest "
                          "(encode_upload_s) -> recursion forest + PRG text on the device, copied back (build_s) -> every locus's .bin "
                          "(uint32 stream) and .gfa text + CRC-32s by the batch encoders (encoders_s); the command line's stages without "
                          "its files; all workers in parallel, native stages with CPUs / workers threads each",
                   max_over_workers_s={k_: round(max(p[k_] for p in parts), 3) for k_ in
                                       ("parse_s", "encode_upload_s", "build_s", "encoders_s")},
                   output_bytes=sum(p["out_bytes"] for p in parts))
    for c in conns:
        c.send(("quit", None))
    for pr in procs:
        pr.join(timeout=60)

    # ---- the same measurement with ONE host worker process (a fresh child process; this one's workers have left the device)
    single = None
    if rank == 0 and world == 1 and W > 1 and not args.no_single_worker_leg:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--workers", "1", "--steps", str(max(3, args.steps // 2)), "--warmup", "2",
               "--batch", str(args.batch), "--no-cpu-baseline", "--no-end-to-end", "--no-single-worker-leg", "--no-cli-leg", "--no-deep-leg",
               "--no-shard-projection"]
        try:
            line = subprocess.run(cmd, capture_output=True, text=True, timeout=900, check=True).stdout.strip().splitlines()[-1]
            one = json.loads(line)
            single = dict(value=one["value"], ms_per_step=one["ms_per_step"], steps=one["steps"],
                          fraction_of_value=round(one["value"] / value, 4),
                          exclusive_pass=one["roofline"]["exclusive_pass"], verified_mismatches=one["config"]["verified"]["mismatches"])
        except Exception as err:          # reported, not hidden
            single = dict(error=f"{type(err).__name__}: {err}"[:300])

    # ---- what one GPU of an N-GPU strong-scaling run sees: the job's alignments / N per step, measured on this GPU with one and
    #      with four host workers (fresh child processes).  The driver computes scaling from its own 8-GPU runs; this is the
    #      projection a one-GPU box can make: N x shard rate / value.
    projection = None
    if rank == 0 and world == 1 and W >= 1 and not args.no_shard_projection:
        import subprocess
        projection = dict(note="alignments per step = batch / N on ONE GPU (what a rank of an N-GPU run builds); projected speed-up = "
                               "N x shard rate / value of this line", shards=[])
        for n_gpus in (2, 4, 8):
            shard = args.batch // n_gpus
            entry = dict(n_gpus=n_gpus, alignments_per_step=shard)
            # planned: the passes after the first of a resident batch (capacities = the batch's own previous totals); first_pass: every
            # pass sized from ANOTHER batch's totals with headroom — what a rank that sees its shard once, or a chunk of the command
            # line, gets (--first-pass).  Host shape: this file's default for the shard's size.
            for label, extra in (("planned", []), ("first_pass", ["--first-pass"])):
                cmd = [sys.executable, os.path.abspath(__file__), "--steps", "6", "--warmup", "2", "--batch", str(shard),
                       "--no-cpu-baseline", "--no-end-to-end", "--no-single-worker-leg", "--no-cli-leg", "--no-shard-projection", "--no-deep-leg"] + extra
                try:
                    line = subprocess.run(cmd, capture_output=True, text=True, timeout=900, check=True).stdout.strip().splitlines()[-1]
                    one = json.loads(line)
                    c1 = one["config"]
                    entry[label] = dict(value=one["value"], ms_per_step=one["ms_per_step"], worker_processes=c1["host_worker_processes_per_gpu"],
                                        engines_per_worker=c1["streams_per_worker"], host_waits_per_step=c1["host_waits_per_step"],
                                        calls_per_step=c1["launches_per_step"], plan_misses_per_step=c1["plan_misses_per_step"],
                                        levels_left_to_the_per_step_host_per_step=c1["plan_resumes_per_step"],
                                        verified_mismatches=c1["verified"]["mismatches"],
                                        projected_speedup=round(n_gpus * one["value"] / value, 3))
                except Exception as err:          # reported, not hidden
                    entry[label] = dict(error=f"{type(err).__name__}: {err}"[:300])
            projection["shards"].append(entry)

    # ---- the command line, file -> file: the same alignments as FASTA files on local disk -> .prg.fa, .prg.bin.zip, .prg.gfa.zip,
    #      update_DS.zip (-O a); a fresh process, so the rate includes interpreter + device start-up
    cli = None
    if cli_dir is not None and rank == 0:
        import hashlib as _hl
        import shutil
        import subprocess
        t_cli = args.cli_threads or max(1, min(16, ncpu))
        outp = os.path.join(cli_root, "out", "pan")
        logp = os.path.join(cli_root, "log.txt")
        cmd = [sys.executable, "-m", "make_prg_amd", "from_msa", "-i", cli_dir, "-o", outp, "-t", str(t_cli), "-O", "a", "--log", logp]
        try:
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1800)
            dt_cli = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-400:])
            sizes = {n_: os.path.getsize(os.path.join(cli_root, "out", n_)) for n_ in sorted(os.listdir(os.path.join(cli_root, "out")))}
            # a sample of the written PRGs against the oracle's digests
            bad_cli = checked = 0
            if os.path.exists(DIGESTS):
                from tests.config_c_full import load_digests
                blob = load_digests(DIGESTS)
                with open(outp + ".prg.fa", "rb") as fh:
                    for k, line in enumerate(fh):
                        if k % 2 == 0:
                            name = line[1:-1].decode()
                        elif (k // 2) % 10 == 0:
                            sd = int(name[4:])
                            if sd * 12 + 12 <= len(blob):
                                checked += 1
                                bad_cli += _hl.sha256(line[:-1]).digest()[:8] != blob[12 * sd:12 * sd + 8]
            pipe_s = next((float(l.split(" in ")[1].split("s")[0]) for l in open(logp) if "built and written" in l), None)
            cli = dict(value=round(len(seeds) / dt_cli, 3), unit="MSAs/s", seconds=round(dt_cli, 3), files=len(seeds), threads=t_cli,
                       output_types="a (.prg.fa, .prg.bin.zip, .prg.gfa.zip, update_DS.zip)", pipeline_seconds=pipe_s,
                       region="python -m make_prg_amd from_msa: FASTA files on local disk -> every output file (a fresh process: "
                              "interpreter, torch import and device start-up included; pipeline_seconds = after the input list is read)",
                       output_bytes=sizes, prgs_checked_against_digests=checked, mismatches=int(bad_cli))
            # the same command with -O p (.prg.fa + update_DS.zip): the build-bound rate of the command line — -O a is bound by
            # writing the 9.6 GB .prg.bin.zip
            shutil.rmtree(os.path.join(cli_root, "out"), ignore_errors=True)
            cmd_p = [c_ if c_ != "a" else "p" for c_ in cmd]
            t0 = time.perf_counter()
            r = subprocess.run(cmd_p, cwd=ROOT, capture_output=True, text=True, timeout=1800)
            dt_p = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-400:])
            cli["prg_only"] = dict(value=round(len(seeds) / dt_p, 3), unit="MSAs/s", seconds=round(dt_p, 3), output_types="p (.prg.fa, update_DS.zip)",
                                   output_bytes={n_: os.path.getsize(os.path.join(cli_root, "out", n_)) for n_ in sorted(os.listdir(os.path.join(cli_root, "out")))})
        except Exception as err:          # reported, not hidden
            cli = dict(cli or {}, error=f"{type(err).__name__}: {err}"[:400])
        shutil.rmtree(cli_root, ignore_errors=True)

    # ---- one DEEP alignment (BASELINE config 4's kind of work at the size the parity fixture pins: tests/golden/ddeep.json, 2 000 x 4 000
    #      hierarchical, -N 7: ~10^4 nodes, ~4 000 KMeans fits of up to 817 sequences x 16 354 k-mers): wall of forest + PRG text in a child
    #      process, the PRG's hash against the fixture's (the real reference's PRG)
    deep = None
    if not args.no_deep_leg and rank == 0 and world == 1:
        import subprocess
        import tempfile
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ddeep.json")))
            with tempfile.TemporaryDirectory() as td:
                outj = os.path.join(td, "deep.json")
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "deep_profile.py"), str(gold["S"]), str(gold["C"]), str(gold["N"]),
                                    "--passes", "2"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, MPRG_DEEP_OUT=outj))
                if r.returncode != 0:
                    raise RuntimeError(r.stderr[-400:])
                dj = json.load(open(outj))
            ps = dj["passes"]
            # roofline of the deep leg's dominant entry point (the profiled pass: HIP events on the launch stream): algorithmic bytes
            # 8 D V (iterations + n_init) of the fits it ran / its device time
            top_e = max(ps[0].get("entry_points", [{}]), key=lambda e_: e_.get("ms", 0.0)) if ps[0].get("entry_points") else None
            deep_roof = None
            if top_e and top_e.get("achieved_GBps"):
                deep_roof = dict(bound="hbm", entry_point=top_e["entry_point"], kernel=KERNEL_OF.get(top_e["entry_point"], "k_kmeans_restart_wide + k_kmeans_select_only_list + k_kmeans_predict_list"),
                                 ms=top_e["ms"], launches=top_e["calls"], algorithmic_bytes=top_e["algorithmic_bytes"], achieved=top_e["achieved_GBps"],
                                 peak=HBM_PEAK_GBS, unit="GB/s", frac=round(top_e["achieved_GBps"] / HBM_PEAK_GBS, 6),
                                 traffic=None, traffic_note="FETCH_SIZE / WRITE_SIZE passes of the same command: profiles/r05/deep/",
                                 # the bytes twice: `algorithmic_bytes` (and `frac`) are of the fits the reference's loop reaches; the launches
                                 # that fit EVERY round of a big level at once also fit rounds it never reaches — work the reference does not do
                                 kmeans_bytes_reference_fits=ps[0].get("kmeans_bytes_reference_fits"),
                                 kmeans_bytes_launched_at_once=ps[0].get("kmeans_bytes_launched_at_once"))
            deep = dict(roofline=deep_roof,workload=f"one hierarchical alignment {gold['S']} x {gold['C']} (utils/synthetic.synth_rows_deep seed {gold['seed']}), -N {gold['N']} -L {gold['L']}",
                        seconds=round(min(p_["wall_ms"] for p_ in ps) / 1e3, 3), nodes=ps[-1]["nodes"], levels=ps[-1]["levels"], kmeans_fits=ps[-1]["fits"],
                        prg_identical_to_the_fixture=all(p_["prg_sha256"] == gold["expect"]["prg_sha256"] for p_ in ps),
                        fixture="tests/golden/ddeep.json (PRG confirmed by the REAL reference: 293 s on this container's CPU)",
                        top_entry_points=[(e_["entry_point"], e_["ms"]) for e_ in ps[0].get("entry_points", [])[:4]],
                        region="resident alignment -> recursion forest (per-step host) + PRG text in pinned memory; levels with big clustering "
                               "problems take mprg_kmeans_fit_wide, every round's fit at once where the level holds few problems "
                               "(profiles/r05/deep_alignment.md)")
        except Exception as err:          # reported, not hidden
            deep = dict(error=f"{type(err).__name__}: {err}"[:400])

    # whole-job counters (strong scaling: a rank's workers only saw its shard)
    keys = ("launches", "fits", "cells_all", "cells_clustered", "kmeans_bytes", "syncs", "plan_misses", "plan_resumes")
    cvec = torch.tensor([float(counters.get(k_, 0)) for k_ in keys], dtype=torch.float64, device=device if (dist_backend == "nccl" and grouped) else "cpu")
    if grouped:
        dist.all_reduce(cvec, op=dist.ReduceOp.SUM)
    for k_, v_ in zip(keys, cvec.tolist()):
        counters[k_] = v_

    if rank == 0:
        prof = excl["prof"] or {"mprg_kmeans_restarts": dict(calls=0, ms=0.0, bytes=0.0)}
        dev_ms = sum(v["ms"] for v in prof.values())
        ranked = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])

        def kern(name, d):
            gbps = (d["bytes"] / max(d["ms"], 1e-9)) * 1e-6 if d["bytes"] else None          # bytes/ms -> GB/s
            return dict(entry_point=name, kernel=KERNEL_OF.get(name, name.replace("mprg_", "k_")), ms=round(d["ms"], 3),
                        launches=d["calls"], share_of_device_time=round(d["ms"] / max(dev_ms, 1e-9), 4),
                        avg_launch_ms=round(d["ms"] / max(d["calls"], 1), 4),
                        algorithmic_bytes_per_launch=round(d["bytes"] / max(d["calls"], 1), 1) if d["bytes"] else None,
                        achieved_GBps=round(gbps, 3) if gbps else None, frac=round(gbps / HBM_PEAK_GBS, 6) if gbps else None)

        name, d = ranked[0]
        top = kern(name, d)
        # HBM traffic of the dominant kernel: from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
        # tools/summarize_pmc.py) — counters cannot be read inside this process — and only if that summary was made from
        # exactly these kernel sources; otherwise null
        traffic = traffic_note = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r06", "pmc_summary.json")))
            if pm.get("source_digest") == source_digest():
                # the PMC passes profile a different batch (8192 alignments in one process), so their bytes per launch are
                # not this pass's: what carries over is HBM bytes / algorithmic bytes of the entry point, measured there
                ratio = pm["entry_points"][name]["traffic_over_algorithmic"]
                traffic = round(ratio * top["algorithmic_bytes_per_launch"], 1)
                traffic_note = (f"{ratio} x algorithmic bytes: (2 x FETCH_SIZE + WRITE_SIZE) / algorithmic bytes of {name} in the PMC "
                                f"passes of the same sources (profiles/r06/pmc_summary.json, source_digest {pm['source_digest']})")
        except Exception:
            pass
        roof = dict(bound="hbm", kernel=top["kernel"], entry_point=name, achieved=top["achieved_GBps"] or 0.0,
                    peak=HBM_PEAK_GBS, unit="GB/s", frac=top["frac"] or 0.0, traffic=traffic, traffic_note=traffic_note,
                    avg_launch_ms=top["avg_launch_ms"], launches=top["launches"],
                    algorithmic_bytes_per_launch=top["algorithmic_bytes_per_launch"],
                    measured="exclusive pass: one worker process alone on the device, one stream, "
                             f"{excl['loci']} alignments, HIP events around every entry point on the launch stream",
                    exclusive_pass=dict(wall_ms=round(excl["wall_ms"], 3), device_ms=round(dev_ms, 3),
                                        wall_over_device=round(excl["wall_ms"] / max(dev_ms, 1e-9), 3),
                                        kmeans_fits=excl["counters"]["fits"], launches=excl["counters"]["launches"],
                                        host_waits=excl["counters"].get("syncs", 0)),
                    kernels=[kern(n_, d_) for n_, d_ in ranked[:6]])
        out = {
            "metric": "MSAs/sec (from_msa, whole node) on 30k-gene pan-genome",
            "value": round(value, 3), "unit": "MSAs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000.0 * dt_max / args.steps, 3), "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None, "dtype": "u8 (+f64 KMeans)", "data": "synthetic",
            "config": {"workload": "C: 30k-gene pan-genome shape (S~N(100,20) in [20,300] rows x 1000-3000 cols, "
                                   "SURVEY.md §8d generator, seeds 0..batch-1), -N 5 -L 7; one step = every alignment of the job, "
                                   "resident, sharded over the ranks by size (--weak: all of them on every rank)",
                       "alignments_per_step": msas_per_step, "alignments_rank0": len(seeds), "parallelism": f"shard{world}",
                       "process_group": (f"{dist.get_backend()} (world {world}" + (", forced: MPRG_DIST_FORCE" if world == 1 else "") + ")") if grouped else None,
                       "host_worker_processes_per_gpu": W, "cpus_rank0": ncpu, "shard_projection": projection, "km_side_streams": os.environ.get("MPRG_KM_SIDE_STREAMS", "0") != "0", "single_worker": single, "cli": cli, "deep_alignment": deep,
                       "streams_per_worker": args.streams, "event_timing_in_timed_region": bool(args.profile_timed),
                       "step_includes": "recursion forest (kernels + device-side bookkeeping; the host sizes buffers from one header per "
                                        "step) + PRG text laid out and written on the device + its copy to pinned host memory",
                       "host_waits_per_step": counters.get("syncs", 0) / args.steps,
                       "first_pass": bool(args.first_pass), "plan_misses_per_step": counters.get("plan_misses", 0) / args.steps,
                       "plan_resumes_per_step": counters.get("plan_resumes", 0) / args.steps,
                       "ingest_s_excluded": round(t_ing, 3), "loci_built_last_step": n_ok,
                       "levels": counters["levels"] / args.steps, "launches_per_step": counters["launches"] / args.steps,
                       "kmeans_fits_per_step": counters["fits"] / args.steps,
                       "B_alg_bytes_per_step": (counters["cells_all"] + counters["cells_clustered"]
                                                + counters["kmeans_bytes"]) / args.steps,
                       "whole_step_GBps": round((counters["cells_all"] + counters["cells_clustered"]
                                                 + counters["kmeans_bytes"]) / args.steps / (dt_max / args.steps) * 1e-9, 3),
                       "verified": verified, "end_to_end": e2e},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if th is not None:
        th.join(timeout=30)
    if grouped:
        dist.destroy_process_group()
    if verified is not None and (verified["mismatches"] or verified.get("mismatches_all_ranks", 0)):
        sys.exit(3)


if __name__ == "__main__":
    main()

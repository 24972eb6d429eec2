"""`from_msa` sub-command: same flags and output files as make_prg/subcommands/from_msa.py:18-198.

Instead of one worker process per alignment, the input list is sharded over the ranks of the job (one process per
GPU; a single process when run plainly) and each rank streams its shard through the file -> file pipeline on its device
(make_prg_amd/pipeline.py).  Per-locus policy is the reference's: a SequenceCurationError skips the locus with a
warning, an empty alignment aborts the run (EmptyMSAError).  Under several ranks every rank writes segment files, rank 0 gathers
the segments' INDEX (the job's one collective) and merges their byte ranges into <prefix>.prg.fa (loci in sorted order),
<prefix>.prg.bin(.zip), <prefix>.prg.gfa(.zip), <prefix>.update_DS.zip (utils/segments.py).
"""
import logging
import os
import pickle
from pathlib import Path
from typing import Dict, List

import numpy as np

from ..engine import BatchEngine, SequenceCurationError, build_prg
from ..from_msa import MIN_MATCH_LEN, NESTING_LVL
from ..msa import load_alignment_file
from ..utils.gfa import GFA_Output
from ..utils.io_utils import output_files_already_exist, remove_known_input_extensions, zip_bytes
from ..utils.prg_encoder import PrgEncoder

logger = logging.getLogger("make_prg_amd")


class EmptyMSAError(Exception):
    pass


def register_parser(subparsers):
    p = subparsers.add_parser("from_msa", usage="make_prg from_msa", help="Make PRG from multiple sequence alignment")
    p.add_argument("-i", "--input", action="store", type=str, required=True,
                   help="Multiple sequence alignment file or a directory containing such files")
    p.add_argument("-s", "--suffix", action="store", type=str, default="",
                   help="If the input parameter (-i, --input) is a directory, then filter for files with this suffix. "
                        "If this parameter is not given, all files in the input directory is considered.")
    p.add_argument("-o", "--output-prefix", dest="output_prefix", action="store", type=str, required=True,
                   help="Prefix for the output files")
    p.add_argument("-f", "--alignment-format", dest="alignment_format", action="store", default="fasta",
                   help="Alignment format of MSA. fasta (default: %(default)s) is read exactly as the reference reads it (every golden file of "
                        "the reference is FASTA).  clustal, stockholm, phylip, phylip-sequential, phylip-relaxed are BEST-EFFORT restatements of "
                        "Biopython's readers (not pinned against Bio.AlignIO: record ids of unusual files may differ); any other AlignIO format "
                        "is an error")
    p.add_argument("-N", "--max-nesting", dest="max_nesting", action="store", type=int, default=NESTING_LVL,
                   help="Maximum number of levels to use for nesting. Default: %(default)d")
    p.add_argument("-L", "--min-match-length", dest="min_match_length", action="store", type=int, default=MIN_MATCH_LEN,
                   help="Minimum number of consecutive characters which must be identical for a match. Default: %(default)d")
    p.set_defaults(func=run)
    return p


def get_all_input_files(input_path: str, suffix: str) -> List[Path]:
    path = Path(input_path)
    if not path.exists():
        raise FileNotFoundError(f"{path} does not exist")
    if path.is_file():
        return [path]
    root = path.resolve()
    with os.scandir(root) as it:          # (one directory read; Path.resolve() / is_file() per entry cost 0.5 s per 30 000 files)
        return [root / e.name for e in it if e.name.endswith(suffix) and e.is_file()]


def _dist():
    """(rank, world, dist module or None): a torch.distributed job if the launcher set one up."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # MPRG_DIST_FORCE=1: a process group (and the multi-rank path: segments, index all-gather, placement) also for ONE rank — what a box
    # with one GPU can show of the multi-GPU job: the library's kernels, RCCL's communicator and one HIP runtime in one process
    if world == 1 and (os.environ.get("MPRG_DIST_FORCE", "0") == "0" or "MASTER_ADDR" not in os.environ):
        return 0, 1, None
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        backend = os.environ.get("MPRG_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend)
        _dist.we_initialised = True
    return dist.get_rank(), dist.get_world_size(), dist


_dist.we_initialised = False


def balanced_parts(files: List[Path], n_parts: int) -> List[List[Path]]:
    """Cost-balanced split: longest-processing-time greedy on file size (gene sizes are skewed), every file stat-ed once,
    one pass with a heap of part loads.  Deterministic (ties by path and part number)."""
    import heapq
    sized = sorted(((-f.stat().st_size, str(f), f) for f in files))
    heap = [(0, p) for p in range(n_parts)]
    parts: List[List[Path]] = [[] for _ in range(n_parts)]
    for neg, _, f in sized:
        load, p = heapq.heappop(heap)
        parts[p].append(f)
        heapq.heappush(heap, (load - neg, p))
    return parts


def shard_files(files: List[Path], rank: int, world: int) -> List[Path]:
    """This rank's shard of the input list: a CONTIGUOUS stretch of the run's sorted loci (the order of every output file,
    utils/input_output_files.py:89), cut where the cumulative file size crosses rank / world of the total — so a rank's part of every
    output file is one byte range (utils/segments.py: every rank places its own bytes), and the shards are balanced by size to
    within one file (a 30 000-gene pan-genome: thousands of files per rank)."""
    from ..pipeline import sort_key
    ordered = sorted(files, key=sort_key)
    sizes = np.asarray([f.stat().st_size for f in ordered], np.float64)
    total = float(sizes.sum())
    if total <= 0:
        return ordered[rank::world]
    mid = np.cumsum(sizes) - sizes / 2.0                   # a file belongs to the rank its midpoint falls into
    owner = np.minimum((mid * world / total).astype(np.int64), world - 1)
    return [f for f, o in zip(ordered, owner.tolist()) if o == rank]


def _load_one(args):
    path, fmt = args
    try:
        return load_alignment_file(path, fmt, defer_n=True)      # N columns are counted on the device, batch-wise
    except ValueError as err:
        return err


def load_many(files: List[Path], alignment_format: str, procs: int = 1) -> list:
    """Parse + upper-case + N-replace every file (utils/io_utils.py:17-49), with `procs` worker processes.
    Must run before the GPU is initialised when procs > 1 (the workers are forked)."""
    jobs = [(f, alignment_format) for f in files]
    if procs > 1 and len(files) >= 32:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(procs) as pool:
            return pool.map(_load_one, jobs, chunksize=16)
    return [_load_one(j) for j in jobs]


BATCH = int(os.environ.get("MPRG_BATCH", "4096"))          # alignments per resident batch
_CHECK_TREES = os.environ.get("MPRG_CHECK", "") not in ("", "0")


def build_shard(files: List[Path], options, backend=None) -> Dict[str, dict]:
    """All loci of this rank: {locus: {prg, bin, gfa, pickle}}; loci skipped by the curation policy are absent.
    Ingest runs in `-t` worker processes (the reference's -t starts that many per-alignment workers); the build runs
    in resident batches of MPRG_BATCH alignments on this rank's GPU."""
    loaded = load_many(files, options.alignment_format, max(1, int(getattr(options, "threads", 1) or 1)))
    out: Dict[str, dict] = {}
    for lo in range(0, len(files), BATCH):
        chunk_files = files[lo:lo + BATCH]
        msas, loci = [], []
        for f, m in zip(chunk_files, loaded[lo:lo + BATCH]):
            locus = remove_known_input_extensions(f.name)
            if isinstance(m, ValueError):
                if "No records found in handle" in str(m.args[0]):
                    raise EmptyMSAError(f"No records found in MSA of locus {locus}")
                raise m
            msas.append(m)
            loci.append(locus)
        _build_batch(msas, loci, options, backend, out)
    return out


def locus_record(locus: str, prg: str, builder, output_type) -> dict:
    """The per-locus output bytes of a run: PRG text, pickled builder (update_DS member), binary PRG, GFA — what the
    reference writes to per-locus temp files (subcommands/from_msa.py:114-133, subcommands/update.py:128-141)."""
    rec = dict(prg=prg)
    if output_type.prg:
        rec["pickle"] = pickle.dumps(builder, protocol=4)
    if output_type.binary:
        enc = PrgEncoder()
        arr = enc.encode_array(prg)
        rec["bin"] = (arr if arr is not None else np.asarray(enc.encode(prg))).astype("<u4").tobytes()
    if output_type.gfa:
        rec["gfa"] = GFA_Output.gfa_bytes(prg)
    return rec


def _build_batch(msas, loci, options, backend, out: Dict[str, dict], ring: int = 0):
    """ring: the set of pinned download buffers of the backend this batch's copies use (forest.assemble_prgs)."""
    from ..device import get_backend
    from ..prg_builder import PrgBuilder
    from ..recursion_tree import materialise
    if not msas:
        return
    be = backend or get_backend()
    ot = options.output_type

    def new_builder(locus, root_factory):
        return PrgBuilder(locus, None, options.alignment_format, options.max_nesting, options.min_match_length,
                          _root_factory=root_factory)

    # the array-at-a-time host assumes unique row ids inside an alignment; the rest keeps the id-based node host
    is_uniq = [len(set(m.ids)) == len(m.ids) for m in msas]
    uniq = [i for i, u in enumerate(is_uniq) if u]
    dup = [i for i, u in enumerate(is_uniq) if not u]
    if uniq:
        from ..forest import ForestEngine
        from ..recursion_tree import materialise_forest
        eng = ForestEngine(be, options.max_nesting, options.min_match_length)
        eng.load([msas[i] for i in uniq])
        eng.run_forest()
        prgs = eng.assemble_prgs(want_index=ot.prg, ring=ring)
        for j, i in enumerate(uniq):
            if prgs[j] is None:
                err = eng.errors[j]
                if not isinstance(err, SequenceCurationError):
                    raise err
                logger.warning(f"Skipping building PRG for {loci[i]}. Error: {err}")
                continue
            logger.info(f"Writing output files of locus {loci[i]}")
            builder = None
            if ot.prg:
                # the pickled builder must be what the reference serialises AFTER build_prg() (subcommands/from_msa.py:
                # 123-127): site counter advanced, every leaf allele in prg_index and in its leaf's indexed intervals —
                # `update` looks leaves up by these keys.  They come from the batch's index arrays, not from a second
                # traversal of the node objects.
                leaf_of = {}
                builder = new_builder(loci[i], lambda b, j=j, i=i: materialise_forest(eng, j, msas[i], b, leaf_of))
                builder.site_num = 5 + 2 * int(eng.site_count[j])
                for a, e, nid in eng.prg_index_entries(j).tolist():
                    builder.update_PRG_index(a, e, leaf_of[nid])
                if _CHECK_TREES:          # MPRG_CHECK=1: re-derive everything from the node objects (slow; tests do)
                    index, site = dict(builder.prg_index), builder.site_num
                    builder.clear_PRG_index()
                    assert builder.build_prg() == prgs[j] and builder.prg_index == index and builder.site_num == site
            out[loci[i]] = locus_record(loci[i], prgs[j], builder, ot)
    if dup:
        eng2 = BatchEngine(be, options.max_nesting, options.min_match_length)
        results = eng2.build([msas[i] for i in dup])
        for i, res in zip(dup, results):
            if res.error is not None:
                if not isinstance(res.error, SequenceCurationError):
                    raise res.error
                logger.warning(f"Skipping building PRG for {loci[i]}. Error: {res.error}")
                continue
            prg, _, _ = build_prg(eng2, res)
            builder = None
            if ot.prg:
                builder = new_builder(loci[i], lambda b, res=res, i=i: materialise(eng2, res, msas[i], b, None))
                assert builder.build_prg() == prg          # fills prg_index / site_num as the reference's build_prg() does
            out[loci[i]] = locus_record(loci[i], prg, builder, ot)


def _build_part(job):
    """One host worker process of `-t`: its share of the rank's files, built on the rank's GPU."""
    files, options = job
    options.threads = 1
    return build_shard(files, options)


def split_for_workers(files: List[Path], n: int) -> List[List[Path]]:
    """Size-balanced split of a rank's files over its host worker processes (same greedy rule as shard_files)."""
    return [p for p in balanced_parts(files, n) if p]


def _write_one(all_loci: Dict[str, dict], kind: str, output_prefix: str):
    single = len(all_loci) == 1
    if kind == "fa":
        with open(output_prefix + ".prg.fa", "w") as fh:
            for locus in sorted(all_loci, key=lambda l: l + ".prg.fa"):       # sorted temp-file paths in the reference
                fh.write(f">{locus}\n{all_loci[locus]['prg']}\n")
    elif kind == "pickle":
        zip_bytes(Path(output_prefix + ".update_DS.zip"), {l: all_loci[l]["pickle"] for l in all_loci})
    elif single:
        with open(f"{output_prefix}.prg.{kind}", "wb") as fh:
            fh.write(next(iter(all_loci.values()))[kind])
    else:
        zip_bytes(Path(f"{output_prefix}.prg.{kind}.zip"), {f"{l}.{kind}": all_loci[l][kind] for l in all_loci})


def write_final_files(all_loci: Dict[str, dict], output_type, output_prefix: str, parallel: bool = False):
    """reference utils/input_output_files.py:73-162.  parallel: one writer THREAD per output file — the containers are
    independent and a stored (uncompressed) zip member is a CRC-32 plus a write, both of which release the GIL; threads
    (not forked processes) so that a rank that has initialised the GPU can use them too."""
    kinds = (["fa", "pickle"] if output_type.prg else []) + (["bin"] if output_type.binary else []) + \
            (["gfa"] if output_type.gfa else [])
    if parallel and len(kinds) > 1 and len(all_loci) >= 256:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(len(kinds)) as pool:
            list(pool.map(lambda k: _write_one(all_loci, k, output_prefix), kinds))
    else:
        for k in kinds:
            _write_one(all_loci, k, output_prefix)


# ---- the job's single exchange: every rank's segment INDEX to rank 0
GATHER_ROUND_BYTES = int(os.environ.get("MPRG_GATHER_ROUND_BYTES", str(1 << 30)))


def allgather_bytes(payload: bytes, dist, rank: int, world: int):
    """all_gather of the ranks' byte counts, then of the payloads as uint8 (RCCL moves device tensors, gloo host tensors), in rounds
    of at most GATHER_ROUND_BYTES per rank.  Returns the list of the ranks' payloads, on every rank.
    What travels is each rank's segment index (loci, record lengths, members' CRC / size / offset: ~100 bytes per locus) — the
    outputs themselves stay in the files the ranks wrote (utils/segments.py)."""
    import torch
    on_device = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
    payload = np.frombuffer(payload, dtype=np.uint8)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([payload.size], dtype=torch.int64, device=dev))
    sizes = [int(t.item()) for t in sizes]
    cap = max(max(sizes), 1)
    host = [np.empty(n, np.uint8) for n in sizes]
    for lo in range(0, cap, GATHER_ROUND_BYTES):
        n = min(GATHER_ROUND_BYTES, cap - lo)
        mine = torch.zeros(n, dtype=torch.uint8, device=dev)
        seg = payload[lo:lo + n]
        if seg.size:
            mine[:seg.size] = torch.from_numpy(seg.copy()).to(dev)
        parts = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(parts, mine)
        for r, t in enumerate(parts):
            take = max(0, min(sizes[r] - lo, n))
            if take:
                host[r][lo:lo + take] = t[:take].cpu().numpy()
        del parts, mine
    return [h.tobytes() for h in host]


def hip_runtimes_mapped() -> List[str]:
    """The distinct HIP runtime libraries mapped into this process (/proc/self/maps).  A rank runs the library's kernels AND
    torch.distributed (RCCL): both must sit on ONE copy of the runtime — two copies mean two sets of streams, devices and contexts
    that know nothing of each other."""
    seen = set()
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                if "libamdhip64" in os.path.basename(path):
                    seen.add(os.path.realpath(path))
    except OSError:
        pass
    return sorted(seen)


SHM_FRACTION = float(os.environ.get("MPRG_SEGMENT_SHM_FRACTION", "0.3"))          # of what /dev/shm and MemAvailable leave, all local ranks together
_SEGMENT_DIRS: List[str] = []                         # private in-memory directories of this process (removed at exit and on SIGTERM / SIGINT)


def _remove_segment_dirs():
    import shutil
    while _SEGMENT_DIRS:
        shutil.rmtree(_SEGMENT_DIRS.pop(), ignore_errors=True)


def _sweep_stale_segment_dirs(root: str = "/dev/shm"):
    """Directories of earlier runs of this user whose process is gone (a rank that was killed outright): their files are memory."""
    import shutil
    tag = f"mprg_{os.getuid()}_"
    try:
        names = [n for n in os.listdir(root) if n.startswith(tag)]
    except OSError:
        return
    for name in names:
        parts = name.split("_")
        try:
            pid = int(parts[2])
            os.kill(pid, 0)                           # (raises when no such process)
        except (IndexError, ValueError):
            continue
        except ProcessLookupError:
            path = os.path.join(root, name)
            if os.path.isdir(path) and not os.path.islink(path) and os.stat(path).st_uid == os.getuid():
                shutil.rmtree(path, ignore_errors=True)
        except PermissionError:
            pass


def _private_shm_dir() -> str:
    """A directory of this process alone in /dev/shm (mode 0700, name unpredictable: mkdtemp), removed when the process ends —
    normally, by an exception, or by the SIGTERM torchrun sends the peers of a rank that failed."""
    import atexit
    import signal
    import tempfile
    _sweep_stale_segment_dirs()
    path = tempfile.mkdtemp(prefix=f"mprg_{os.getuid()}_{os.getpid()}_", dir="/dev/shm")
    if not _SEGMENT_DIRS:
        atexit.register(_remove_segment_dirs)
        for sig in (signal.SIGTERM, signal.SIGINT):
            prev = signal.getsignal(sig)

            def handler(signum, frame, prev=prev):
                _remove_segment_dirs()
                if callable(prev):
                    prev(signum, frame)
                else:
                    signal.signal(signum, signal.SIG_DFL)
                    os.kill(os.getpid(), signum)
            try:
                signal.signal(sig, handler)
            except ValueError:                        # (not the main thread: atexit still runs)
                pass
    _SEGMENT_DIRS.append(path)
    return path


def segment_prefix(options, rank: int, files: List[Path]) -> str:
    """Where this rank writes its segment files: in memory when the shard's outputs fit comfortably (they are ~4x the inputs with
    -O a) — a PRIVATE directory under /dev/shm (_private_shm_dir), at most SHM_FRACTION of what is free there and in memory for all
    local ranks together —, else next to the run's output.  MPRG_SEGMENT_DIR overrides (MPRG_SEGMENT_DIR=/dev/shm forces the
    in-memory directory, any other value is used as it is)."""
    base = f"{os.path.basename(options.output_prefix)}.rank{rank}"
    seg_dir = os.environ.get("MPRG_SEGMENT_DIR")
    if seg_dir is None:
        seg_dir = os.path.dirname(os.path.abspath(options.output_prefix))
        try:
            need = 6 * sum(f.stat().st_size for f in files)
            st = os.statvfs("/dev/shm")
            avail_kib = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1])
            world_local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
            if need * world_local < SHM_FRACTION * min(st.f_bavail * st.f_frsize, avail_kib * 1024):
                seg_dir = "/dev/shm"
        except Exception:
            pass
    if seg_dir == "/dev/shm":
        seg_dir = _private_shm_dir()
    return os.path.join(seg_dir, base)


def run_ranks(mine: List[Path], options, backend, dist, rank: int, world: int) -> int:
    """A rank of a multi-GPU run: its shard — a contiguous stretch of the run's sorted loci — through the streamed one-GPU pipeline
    into SEGMENT files of its own; the segment indexes all-gathered (the job's one collective); then every rank copies ITS bytes to
    their final offsets in the run's files, which rank 0 created and closes with the central directories (utils/segments.py) — in
    the reference's order: every output lists the loci sorted (utils/input_output_files.py:89).  Returns the number of loci built."""
    import copy
    import time
    from ..device import get_backend
    from ..pipeline import run_pipeline
    from ..utils import segments
    opts = copy.copy(options)
    opts.output_prefix = segment_prefix(options, rank, mine)
    t0 = time.perf_counter()

    def agree(ok: bool) -> bool:
        """AND over the ranks (a MIN all-reduce): a rank that failed on its own tells the others before the next collective."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    try:
        idx, err = None, None
        try:
            idx = run_pipeline(mine, opts, backend or (lambda: get_backend("runtime")), segment=True)
            libs = hip_runtimes_mapped()
            if len(libs) > 1:
                raise RuntimeError("two HIP runtimes are mapped into this rank (" + ", ".join(libs) + "): the kernels' library and "
                                   "torch.distributed must share one — import torch before the backend is made")
        except Exception as e:          # noqa: BLE001 — told to the other ranks first
            err = e
        if not agree(err is None):
            raise err if err is not None else RuntimeError(f"rank {rank}: another rank failed while building its shard")
        t1 = time.perf_counter()
        got = allgather_bytes(segments.pack_index(idx), dist, rank, world)
        t2 = time.perf_counter()
        n = segments.merge_segments([segments.unpack_index(b) for b in got], options.output_prefix,
                                    sort_key=lambda locus: locus + ".prg.fa", rank=rank, world=world, barrier=dist.barrier, agree=agree)
        t3 = time.perf_counter()
        logger.info(f"rank {rank}: segments built and written in {t1 - t0:.2f}s, index exchange {t2 - t1:.2f}s, placed in the run's files in {t3 - t2:.2f}s")
        run_ranks.timings = dict(build_write_s=t1 - t0, exchange_s=t2 - t1, place_s=t3 - t2)
    finally:
        for suffix in (".prg.fa", ".prg.bin.zip", ".prg.gfa.zip", ".update_DS.zip"):          # (an error on the way: no stray segment)
            try:
                os.remove(opts.output_prefix + suffix)
            except OSError:
                pass
        _remove_segment_dirs()
    return n


def run(cl_options, backend=None):
    options = cl_options
    # `-t N`: the reference starts N per-alignment worker processes; here N host worker processes share the rank's GPU,
    # each building its part of the shard in resident batches (the array-at-a-time host control of one batch overlaps
    # the kernels of the others).  They are forked before this process touches the GPU (a forked HIP context is
    # unusable); with an explicit backend (tests) everything runs in-process.
    n_workers = max(1, int(getattr(options, "threads", 1) or 1)) if backend is None else 1
    pool = None
    # the streamed pipeline (one GPU, or every rank of several) runs its host stages on `-t` THREADS of this process; only the
    # object path (MPRG_PIPELINE=0, one rank) forks worker processes
    pipeline = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("MPRG_PIPELINE", "1") != "0"
    if n_workers > 1 and not pipeline:
        import multiprocessing as mp
        pool = mp.get_context("fork").Pool(n_workers)
    try:
        return _run(options, backend, pool, n_workers)
    finally:
        if pool is not None:
            pool.close()
            pool.join()


def _run(options, backend, pool, n_workers):
    rank, world, dist = _dist()
    input_files = get_all_input_files(options.input, options.suffix)
    if len(input_files) == 0:
        raise FileNotFoundError(f"No input files found in {options.input}")
    if not options.force and output_files_already_exist(options.output_type, options.output_prefix):
        raise RuntimeError("One or more output files already exists, aborting run...")
    Path(options.output_prefix).parent.mkdir(parents=True, exist_ok=True)
    mine = shard_files(input_files, rank, world) if world > 1 else input_files
    import time
    t0 = time.time()
    if dist is None and len(mine) > 1 and os.environ.get("MPRG_PIPELINE", "1") != "0":
        # one GPU: the streamed file -> file pipeline (native batch parser and encoders with `-t` threads, chunks of
        # alignments resident on the device, containers written while the next chunk builds); make_prg_amd/pipeline.py
        from ..device import get_backend
        from ..pipeline import run_pipeline
        n_built = run_pipeline(mine, options, backend or (lambda: get_backend("runtime")))
        logger.info(f"{n_built} of {len(mine)} loci built and written in {time.time() - t0:.1f}s ({max(1, int(getattr(options, 'threads', 1) or 1))} host threads)")
        if n_built == 0:
            logger.error("No PRGs were built, please check errors")
        return
    if dist is not None:
        # several GPUs: every rank streams its shard into segment files, rank 0 merges them by an index (run_ranks)
        import torch
        n_built = run_ranks(mine, options, backend, dist, rank, world)
        logger.info(f"rank {rank}: shard of {len(mine)} loci built and written in {time.time() - t0:.1f}s")
        if dist.get_backend() == "nccl":
            torch.cuda.synchronize()
        dist.barrier()          # (the segments are merged and removed: every rank may leave)
        if rank == 0:
            logger.info(f"{n_built} loci in the merged output files after {time.time() - t0:.1f}s ({world} ranks)")
            if n_built == 0:
                logger.error("No PRGs were built, please check errors")
        if _dist.we_initialised:
            dist.destroy_process_group()
        return
    if pool is not None and len(mine) >= 2 * n_workers:
        # several parts per worker, collected as they finish: the writer-side unpickling of one part overlaps the
        # building of the others (the per-locus outputs are ~0.5 MB each)
        n_parts = max(n_workers, min(4 * n_workers, len(mine) // 256))
        local = {}
        for part in pool.imap_unordered(_build_part, [(p, options) for p in split_for_workers(mine, n_parts)]):
            local.update(part)
    else:
        local = build_shard(mine, options, backend)
    logger.info(f"rank {rank}: {len(local)} of {len(mine)} loci built in {time.time() - t0:.1f}s ({n_workers} host workers)")
    t0 = time.time()
    if not local:
        logger.error("No PRGs were built, please check errors")
        return
    write_final_files(local, options.output_type, options.output_prefix, parallel=True)
    logger.info(f"output files written in {time.time() - t0:.1f}s")

"""Placing the ranks' output SEGMENTS of a multi-GPU from_msa run into the run's files (reference
utils/input_output_files.py:73-162: per-locus temp files of every worker concatenated — sorted — and zipped at the end).

Every rank streams its shard through the one-GPU pipeline (pipeline.py) into files of its own: `<segment prefix>.prg.fa`, the
containers as STORED zips — in memory (/dev/shm) when the shard's outputs fit, next to the output otherwise.  A stored member is a
local header and the data, contiguous and position-independent; a `.prg.fa` record is two lines.  So the run's files are byte
RANGES of the segments in the run's locus order plus one new central directory per container.  Every rank gets every rank's index
(loci, record lengths, members' CRC / size / offset: ~100 bytes per locus through the job's one collective), so every rank knows
where ITS bytes go: rank 0 creates the files at their final sizes, every rank copies its own ranges to their final offsets side by
side (copy_file_range, falling back to read/write; a rank's shard is a contiguous stretch of the run's sorted loci —
subcommands/from_msa.shard_files — so that is one range per file), rank 0 writes the directories.  No rank copies another rank's
bytes (round 4: rank 0 copied all 22 GB of a 30 000-locus -O a run after the ranks had finished).  The result equals what one
rank writes for the whole input, byte for byte.  Nothing of the payload crosses RCCL or Python."""
import json
import os
import struct
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List

import numpy as np

from .zip_stream import StoredZipWriter

COPY_THREADS = int(os.environ.get("MPRG_MERGE_THREADS", "4"))
_KIND_FILE = {"bin": ".prg.bin.zip", "gfa": ".prg.gfa.zip", "pickle": ".update_DS.zip"}


def pack_index(idx: dict) -> bytes:
    return json.dumps(idx, separators=(",", ":")).encode()


def unpack_index(blob) -> dict:
    return json.loads(bytes(blob).decode())


def _copy_range(src_fd: int, dst_fd: int, src_off: int, dst_off: int, n: int):
    """n bytes from src_fd @ src_off to dst_fd @ dst_off, in the kernel where the file systems allow it."""
    use_cfr = hasattr(os, "copy_file_range")
    while n > 0:
        if use_cfr:
            try:
                done = os.copy_file_range(src_fd, dst_fd, min(n, 1 << 30), src_off, dst_off)
                if done > 0:
                    n -= done; src_off += done; dst_off += done
                    continue
            except OSError:
                pass
            use_cfr = False          # (different file systems, an old kernel, a file system without it)
        buf = os.pread(src_fd, min(n, 8 << 20), src_off)
        if not buf:
            raise OSError("a segment is shorter than its index says")
        done = 0
        while done < len(buf):
            done += os.pwrite(dst_fd, memoryview(buf)[done:], dst_off + done)
        n -= len(buf); src_off += len(buf); dst_off += len(buf)


def _copy_plan(dst_path: str, seg_paths: List[str], seg: np.ndarray, src_off: np.ndarray, length: np.ndarray, dst_off: np.ndarray,
               only_rank=None):
    """Copies piece q = length[q] bytes of segment seg[q] @ src_off[q] to dst_off[q]; neighbouring pieces of one segment that are
    neighbours in the destination too go as one range; a few threads share the ranges.  only_rank: the pieces of that segment
    only (every rank places its own bytes)."""
    if only_rank is not None:
        sel = seg == only_rank
        seg, src_off, length, dst_off = seg[sel], src_off[sel], length[sel], dst_off[sel]
    n = len(seg)
    if n == 0:
        return
    cont = (seg[1:] == seg[:-1]) & (src_off[1:] == src_off[:-1] + length[:-1]) & (dst_off[1:] == dst_off[:-1] + length[:-1])
    first = np.concatenate([[True], ~cont])
    run_id = np.cumsum(first) - 1
    starts = np.nonzero(first)[0]
    run_len = np.bincount(run_id, weights=length).astype(np.int64)
    runs = []
    for s_, so, do, ln in zip(seg[starts].tolist(), src_off[starts].tolist(), dst_off[starts].tolist(), run_len.tolist()):
        for lo in range(0, max(ln, 1), 1 << 28):          # (a rank's whole container is one run: share it among the threads)
            runs.append((s_, so + lo, do + lo, min(1 << 28, ln - lo)))
    dst_fd = os.open(dst_path, os.O_WRONLY)
    used = set(seg.tolist())          # (a rank without files wrote no segment)
    fds = [os.open(p, os.O_RDONLY) if r in used else -1 for r, p in enumerate(seg_paths)]
    try:
        def work(part):
            for s_, so, do, ln in part:
                _copy_range(fds[s_], dst_fd, so, do, ln)
        k = max(1, min(COPY_THREADS, len(runs)))
        with ThreadPoolExecutor(k) as pool:
            list(pool.map(work, [runs[q::k] for q in range(k)]))
    finally:
        os.close(dst_fd)
        for fd in fds:
            if fd >= 0:
                os.close(fd)


def merge_segments(indexes: List[dict], output_prefix: str, sort_key, keep_segments: bool = False, rank: int = 0, world: int = 1,
                   barrier=None, agree=None) -> int:
    """indexes[r]: rank r's segment_index (pipeline.py).  Writes <prefix>.prg.fa, .prg.bin(.zip), .prg.gfa(.zip), .update_DS.zip in
    the run's locus order (sort_key(locus)) and removes the segments.  Returns the number of loci.
    world > 1: called by EVERY rank with the same indexes; rank 0 creates the files and writes the directories, every rank copies its
    own segment's ranges (barrier(): the job's barrier, between the three steps).  world == 1: one caller does it all.
    agree(ok) (optional, world > 1): the job's AND over the ranks' `ok` (a MIN all-reduce) — asked after the files are created and after
    the copies, so that a rank that failed on its own (no space, no permission) makes EVERY rank raise instead of leaving the others in the
    next collective until its timeout; rank 0 then removes the half-filled outputs."""
    solo = world == 1
    only = None if solo else rank
    sync = barrier if (barrier is not None and not solo) else (lambda: None)
    created: List[str] = []

    def settle(err, what):
        """Every rank learns whether all of them got through `what`; on a failure anywhere: outputs removed (rank 0), everybody raises."""
        ok = err is None
        all_ok = agree(ok) if (agree is not None and not solo) else ok
        if all_ok:
            return
        if rank == 0:
            for path in created:
                try:
                    os.remove(path)
                except OSError:
                    pass
        if err is not None:
            raise err
        raise RuntimeError(f"another rank failed while {what}: the run's output files were removed")

    def allocate(fd, nbytes):
        """The file at its final size with its blocks reserved (a full disk shows HERE, on rank 0, not in the middle of the ranks' copies)."""
        try:
            if nbytes > 0:
                os.posix_fallocate(fd, 0, nbytes)
            else:
                os.ftruncate(fd, 0)
        except OSError as e:
            import errno
            if e.errno not in (errno.EOPNOTSUPP, errno.EINVAL, errno.ENOSYS):
                raise
            os.ftruncate(fd, nbytes)
    # ---- the run's order
    loci, rank_of, pos_in_rank = [], [], []
    for r, idx in enumerate(indexes):
        for q, (nm, _) in enumerate(idx["fa"] if idx["fa"] else []):
            loci.append(nm); rank_of.append(r); pos_in_rank.append(q)
    have_fa = any(idx["fa"] for idx in indexes)
    kinds = sorted({k for idx in indexes for k in idx["zips"]})
    if not have_fa:          # -O b / g only: the loci come from a container's members
        for r, idx in enumerate(indexes):
            for q, (nm, _, _, _) in enumerate(idx["zips"][kinds[0]] if kinds and kinds[0] in idx["zips"] else []):
                loci.append(nm.rsplit(".", 1)[0] if kinds[0] != "pickle" else nm); rank_of.append(r); pos_in_rank.append(q)
    n_loci = len(loci)
    order = sorted(range(n_loci), key=lambda q: sort_key(loci[q]))
    place = {loci[q]: p for p, q in enumerate(order)}          # locus -> its place in the run
    seg_prefix = [idx["prefix"] for idx in indexes]
    plans = []          # (destination, segment paths, seg, src offsets, lengths, destination offsets)
    writers = []        # rank 0: containers whose central directory is written after the copies
    to_create = []      # rank 0: (path, final size, the container's entries or None for a plain file)
    # ---- <prefix>.prg.fa
    if have_fa:
        src_off = [np.cumsum([0] + [ln for _, ln in idx["fa"]])[:-1] if idx["fa"] else np.zeros(0, np.int64) for idx in indexes]
        seg = np.asarray([rank_of[q] for q in order], np.int64)
        so = np.asarray([src_off[rank_of[q]][pos_in_rank[q]] for q in order], np.int64)
        ln = np.asarray([indexes[rank_of[q]]["fa"][pos_in_rank[q]][1] for q in order], np.int64)
        do = np.cumsum(ln) - ln
        dst = output_prefix + ".prg.fa"
        if rank == 0:
            to_create.append((dst, int(ln.sum()), None))
        plans.append((dst, [p + ".prg.fa" for p in seg_prefix], seg, so, ln, do))
    # ---- the containers: members in the run's order, one new central directory
    for kind in kinds:
        members = []          # (place, rank, name, crc, size, offset in the segment)
        for r, idx in enumerate(indexes):
            for nm, crc, size, off in idx["zips"].get(kind, []):
                locus = nm if kind == "pickle" else nm[:-len(kind) - 1]
                members.append((place.get(locus, n_loci), r, nm, crc, size, off))
        members.sort(key=lambda m: (m[0], m[2]))
        if not members:
            continue
        single = n_loci == 1 and kind != "pickle"          # a single locus is written bare (utils/input_output_files.py:104-131)
        dst = f"{output_prefix}.prg.{kind}" if single else output_prefix + _KIND_FILE[kind]
        head = np.asarray([30 + len(m[2].encode("utf-8")) for m in members], np.int64)
        size = np.asarray([m[4] for m in members], np.int64)
        seg = np.asarray([m[1] for m in members], np.int64)
        so = np.asarray([m[5] for m in members], np.int64)
        if single:
            if rank == 0:
                to_create.append((dst, int(size[0]), None))
            plans.append((dst, [p + _KIND_FILE[kind] for p in seg_prefix], seg, so + head, size, np.zeros(1, np.int64)))
            continue
        ln = head + size
        do = np.cumsum(ln) - ln
        if rank == 0:
            entries = [(m[2].encode("utf-8"), int(m[3]) & 0xFFFFFFFF, int(m[4]), int(o)) for m, o in zip(members, do.tolist())]
            to_create.append((dst, int(ln.sum()), entries))
        plans.append((dst, [p + _KIND_FILE[kind] for p in seg_prefix], seg, so, ln, do))
    err = None
    try:
        for dst, nbytes, entries in to_create:          # (rank 0 only)
            created.append(dst)
            if entries is None:
                with open(dst, "wb") as fh:
                    allocate(fh.fileno(), nbytes)
            else:
                w = StoredZipWriter(dst, threads=1)
                w._open()
                allocate(w.fd, nbytes)
                w.entries = entries
                w.offset = nbytes
                writers.append(w)
    except Exception as e:          # noqa: BLE001 — told to the other ranks below
        err = e
    settle(err, "creating the run's files")
    sync()                # the files exist at their final sizes
    err = None
    try:
        for dst, paths, seg, so, ln, do in plans:
            _copy_plan(dst, paths, seg, so, ln, do, only_rank=only)
    except Exception as e:          # noqa: BLE001
        err = e
    settle(err, "placing the segments' bytes")
    sync()                # every rank's bytes are in place
    for w in writers:
        w.close()
    if not keep_segments:
        for r, p in enumerate(seg_prefix):
            if only is not None and r != only:
                continue
            for suffix in [".prg.fa"] + list(_KIND_FILE.values()):
                try:
                    os.remove(p + suffix)
                except FileNotFoundError:
                    pass
    return n_loci

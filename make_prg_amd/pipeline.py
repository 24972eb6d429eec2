"""The from_msa driver's file -> file pipeline for one GPU (reference subcommands/from_msa.py:136-198:
a process pool over loci, every locus parsed, built, pickled and encoded into temp files that are zipped at the end).

Stages, each working on a CHUNK of a few thousand alignment files and overlapping with the other stages of neighbouring chunks
(three Python threads whose heavy parts — native threads, device waits, file writes — run outside the GIL):
  ingest   libmprg's batch parser (mprg_ingest_*_host, `-t` threads) reads and parses the files straight into a pinned arena
  build    ONE upload, the recursion forest and the PRG text on the device (forest.ForestEngine), text + tree export copied
           back to pinned buffers on the copy stream
  output   libmprg's batch encoders (mprg_encode_*_host, `-t` threads): binary PRG, GFA, CRC-32 of every member; the containers
           are STREAMED (utils/zip_stream.py): members are slices of the chunk's buffers written at their offsets, update_DS
           members are packed from the device's tree export (update_ds.py) — no per-locus temp file, pickle or Python object.
Files the native parser leaves alone (gzip, bytes outside plain ASCII, duplicate row ids, other alignment formats) take the
object path of subcommands/from_msa.py inside their chunk, so the outputs keep the reference's order: loci sorted by name.
"""
import ctypes
import logging
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import List

import numpy as np

from .engine import SequenceCurationError
from .forest import ForestEngine
from .update_ds import member_header
from .utils import native
from .utils.io_utils import remove_known_input_extensions
from .utils.zip_stream import StoredZipWriter

logger = logging.getLogger("make_prg_amd")
CHUNK = int(os.environ.get("MPRG_CHUNK", "1536"))          # alignment files per resident batch (profiles/r03/cli/cli_runtime_backend_chunk_sizes.txt)
DEPTH = 3                                                  # chunks in flight: build | encode | write
TRACE = os.environ.get("MPRG_PIPELINE_TRACE", "") not in ("", "0")
# threads that write one container side by side: buffered writes to ONE file take the inode's lock in turn (tools/write_probe.py:
# 1 / 4 / 16 threads on a file all reach ~10 GB/s), so more than two only spin on that lock
WRITE_THREADS = int(os.environ.get("MPRG_WRITE_THREADS", "2"))
# update_DS members keep their locus's alignment at four bits per cell, packed by the device (update_ds.py "nib4"); 0: ASCII from the parser's arena
PACK_ALIGNMENTS = os.environ.get("MPRG_PACK_ALIGNMENTS", "1") != "0"
TWO_ENGINES = os.environ.get("MPRG_PIPELINE_ENGINES", "2") != "1"          # chunks in turn on two backends (streams): see run_pipeline


def _trace(msg):
    if TRACE:
        import sys
        sys.stderr.write(f"[pipeline] +{since_process_start():6.3f} s  " + msg + "\n")


def since_process_start() -> float:
    """Seconds since this process was created (/proc: start time in clock ticks since boot against the uptime)."""
    with open("/proc/self/stat") as fh:
        ticks = int(fh.read().rsplit(")", 1)[1].split()[19])
    with open("/proc/uptime") as fh:
        up = float(fh.read().split()[0])
    return up - ticks / os.sysconf("SC_CLK_TCK")


def sort_key(path: Path) -> str:
    return remove_known_input_extensions(path.name) + ".prg.fa"          # the reference sorts its per-locus temp paths


class _Outputs:
    """The run's containers, streamed."""

    def __init__(self, prefix: str, ot, threads: int, segment: bool = False):
        self.prefix, self.ot = prefix, ot
        self.segment = segment          # a rank's part of a multi-rank run: containers stay zips, the caller merges them
        self.fa_index = []              # (locus, bytes of its ">locus\nPRG\n" record) in file order
        self.fa_fd = None
        self.zips = {}
        self.threads = threads
        self.n = 0
        self.last = None
        self.pools = {}
        self.lib = None

    def pool(self, lib, slot):
        """The encode pool of a chunk slot (chunks DEPTH apart share one: the earlier one's members are on disk by then)."""
        self.lib = lib
        if slot not in self.pools:
            self.pools[slot] = lib.mprg_encode_pool_new_host()
        else:
            lib.mprg_encode_pool_reset_host(self.pools[slot])
        return self.pools[slot]

    def zip(self, kind):
        if kind not in self.zips:
            name = f"{self.prefix}.update_DS.zip" if kind == "pickle" else f"{self.prefix}.prg.{kind}.zip"
            self.zips[kind] = StoredZipWriter(name, threads=WRITE_THREADS)
        return self.zips[kind]

    def plan_fa(self, addr, ln, lib, threads: int):
        """Places pieces (address table) at the end of <prefix>.prg.fa and returns the function that writes them there."""
        if self.fa_fd is None:
            self.fa_fd = os.open(self.prefix + ".prg.fa", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            self.fa_off = 0
        addr, ln = np.ascontiguousarray(addr, np.int64), np.ascontiguousarray(ln, np.int64)
        off = self.fa_off + np.cumsum(ln) - ln
        self.fa_off += int(ln.sum())
        fd = self.fa_fd

        def run():
            if len(ln) and lib.mprg_write_pieces_host(fd, addr.ctypes.data, ln.ctypes.data, off.ctypes.data, len(ln), threads) != 0:
                raise OSError(f"writing {self.prefix}.prg.fa failed")

        return run

    def close(self):
        for pool in self.pools.values():
            self.lib.mprg_encode_pool_free_host(pool)
        self.pools = {}
        if self.fa_fd is not None:
            os.close(self.fa_fd)
        for z in self.zips.values():
            z.close()
        if self.n == 1 and not self.segment:          # a single locus is written bare, not zipped (utils/input_output_files.py:104-131)
            import zipfile
            for kind in ("bin", "gfa"):
                if kind in self.zips:
                    zpath = f"{self.prefix}.prg.{kind}.zip"
                    with zipfile.ZipFile(zpath) as z:
                        data = z.read(z.namelist()[0])
                    with open(f"{self.prefix}.prg.{kind}", "wb") as fh:
                        fh.write(data)
                    os.remove(zpath)


def _ingest(lib, paths: List[Path], threads: int):
    blob = b"".join(os.fsencode(str(p)) + b"\0" for p in paths)
    h = lib.mprg_ingest_open_host(blob, len(paths), threads)
    info = np.zeros((len(paths), 5), np.int64)
    lib.mprg_ingest_info_host(h, info.ctypes.data)
    return h, info


def run_pipeline(files: List[Path], options, backend, segment: bool = False):
    """Builds every locus of `files` and writes the run's output files.  Returns the number of loci built — or, for a rank's
    SEGMENT of a multi-rank run (segment=True: containers stay zips even for one locus), the segment's index (segment_index).
    
    backend: a backend object, or a function that makes one — it is called AFTER the ingest thread has started, so that reading
    and parsing the first chunks overlaps bringing up the device (0.2 s with the library's own runtime plumbing, 1 s with torch)."""
    from .subcommands import from_msa as drv
    if TRACE:
        import atexit
        _trace(f"pipeline starts {since_process_start():.2f} s after the process ({len(files)} files)")
        atexit.register(lambda: _trace(f"interpreter exits {since_process_start():.2f} s after the process started"))
    lib = native.library()
    if lib is None:
        raise RuntimeError("libmprg_host.so is missing: build it with `python __graft_entry__.py`")
    ot = options.output_type
    threads = max(1, int(getattr(options, "threads", 1) or 1))
    files = sorted(files, key=sort_key)
    chunks = [files[lo:lo + CHUNK] for lo in range(0, len(files), CHUNK)]
    out = _Outputs(options.output_prefix, ot, threads, segment)
    fasta = options.alignment_format == "fasta"
    q_in: "queue.Queue" = queue.Queue(maxsize=2)
    q_out: "queue.Queue" = queue.Queue()
    errors: List[BaseException] = []
    # a chunk's buffers (pinned arena, PRG text, tree export, encode pool: DEPTH of each, chunk i uses slot i % DEPTH) are read by
    # the output stages until its members are written: the build of chunk i + DEPTH starts only when the writes of chunk i are done
    slot_free = [threading.Semaphore(1) for _ in range(DEPTH)]
    writers = ThreadPoolExecutor(int(os.environ.get("MPRG_WRITERS", "4")))

    def stage_ingest():
        try:
            for ci, chunk in enumerate(chunks):
                if errors:
                    break
                import time
                t0 = time.perf_counter()
                if fasta:
                    h, info = _ingest(lib, chunk, threads)
                    _trace(f"chunk {ci}: read + scan {1e3 * (time.perf_counter() - t0):.0f} ms")
                else:
                    h, info = None, np.full((len(chunk), 5), -3, np.int64)
                q_in.put((ci, chunk, h, info))
        except BaseException as err:
            errors.append(err)
        q_in.put(None)

    def stage_output():
        try:
            while True:
                item = q_out.get()
                if item is None:
                    break
                slot = slot_free[item[0]["ci"] % DEPTH]
                jobs = _write_chunk(lib, out, options, threads, *item)
                # the containers are written side by side while the next chunk is encoded; the chunk's buffers are free again
                # once all of them are done
                futs = [writers.submit(j) for j in jobs]
                pending = [len(futs)]
                lock = threading.Lock()

                def done(f, pending=pending, lock=lock, slot=slot):
                    if f.exception() is not None:
                        errors.append(f.exception())
                    with lock:
                        pending[0] -= 1
                        last = pending[0] == 0
                    if last:
                        slot.release()

                if not futs:
                    slot.release()
                for f in futs:
                    f.add_done_callback(done)
        except BaseException as err:
            errors.append(err)
            for sl in slot_free:          # nothing more is written: let the build loop run out
                sl.release()
            while q_out.get() is not None:
                for sl in slot_free:
                    sl.release()

    t_in, t_out = threading.Thread(target=stage_ingest, daemon=True), threading.Thread(target=stage_output, daemon=True)
    t_in.start()
    t_out.start()
    be = backend() if callable(backend) else backend
    _trace(f"device ready {since_process_start():.2f} s after the process started") if TRACE else None
    be.async_depth = DEPTH
    # two backends (streams, pinned rings) take the chunks in turn: chunk i + 1 is uploaded and its forest enqueued BEFORE the host
    # waits for chunk i, lays its text out and hands it to the output stage — the device works on one chunk while the host finishes
    # the other (a backend object handed in — tests — builds chunk by chunk)
    bes = [be]
    if TWO_ENGINES and len(chunks) > 1 and hasattr(be, "clone"):
        bes.append(be.clone())          # (its own streams and device buffers; this backend's pinned host memory)
    closer = ThreadPoolExecutor(1)
    release = lambda h: closer.submit(lib.mprg_ingest_close_host, h)
    pending = None
    try:
        while True:
            if TRACE:
                import time as _t
                t_q = _t.perf_counter()
            item = q_in.get()
            if item is None or errors:
                break
            ci, chunk, h, info = item
            if TRACE:
                import time as _t
                t_w = _t.perf_counter()
            slot_free[ci % DEPTH].acquire()
            if TRACE:
                _trace(f"chunk {ci}: waited {1e3 * (t_w - t_q):.0f} ms for the parser, {1e3 * (_t.perf_counter() - t_w):.0f} ms for its buffers")
            if errors:
                break
            b_ = bes[ci % len(bes)]
            if getattr(b_, "plan_donor", None) is None:          # (the other backend's last chunk sizes this one's first)
                b_.plan_donor = next((x.plan_donor for x in bes if getattr(x, "plan_donor", None) is not None), None)
            st = _build_begin(lib, b_, options, threads, ci, chunk, h, info)
            if len(bes) == 1:
                q_out.put(_build_end(lib, options, threads, st, release))
                continue
            if pending is not None:
                q_out.put(_build_end(lib, options, threads, pending, release))
            pending = st
        if pending is not None and not errors:
            q_out.put(_build_end(lib, options, threads, pending, release))
            pending = None
    finally:
        closer.shutdown(wait=True)
        q_out.put(None)
        t_out.join()
        writers.shutdown(wait=True)
    if errors:
        raise errors[0]
    out.close()
    if os.environ.get("MPRG_FAST_EXIT", "1") == "0":          # (the command line leaves through os._exit instead: un-pinning GBs is slow)
        for b_ in reversed(bes[1:] + ([be] if callable(backend) else [])):          # clones first, then a backend made here
            if hasattr(b_, "close"):
                b_.close()
    _trace(f"outputs closed {since_process_start():.2f} s after the process started") if TRACE else None
    return segment_index(out) if segment else out.n


def segment_index(out: "_Outputs") -> dict:
    """What rank 0 needs to merge a rank's output segment into the run's files without reading it: the loci in file order with
    the length of each .prg.fa record, and per container the members' CRC-32, size and local-header offset."""
    idx = dict(n=out.n, prefix=out.prefix, fa=[[nm, int(ln)] for nm, ln in out.fa_index], zips={})
    for kind, z in out.zips.items():
        idx["zips"][kind] = [[nb.decode("utf-8"), crc, size, off] for nb, crc, size, off in z.entries]
    return idx


def _build_begin(lib, be, options, threads, ci, chunk, h, info):
    """Device stage of one chunk, first half: the fast files parsed into the chunk's pinned arena, ONE upload, device ingest, and the
    whole recursion forest ENQUEUED (forest.forest_enqueue: from the previous chunk's totals without a host wait; the first chunk
    of a run takes the per-step host here).  _build_end() completes it: the two halves of successive chunks are interleaved on two
    backends (streams), so a chunk's kernels run while the host lays out the one before."""
    from .subcommands import from_msa as drv
    import time
    t_start = time.perf_counter()
    ot = options.output_type
    status, rows, cols, tbytes, flags = (info[:, k] for k in range(5))
    for i in np.nonzero((status == -7) | (status == -6))[0].tolist():
        locus = remove_known_input_extensions(chunk[i].name)
        if status[i] == -7:
            raise drv.EmptyMSAError(f"No records found in MSA of locus {locus}")
        raise FileNotFoundError(f"{chunk[i]} could not be read")
    if (status == -5).any():
        raise ValueError("Sequences must all be the same length")
    fast = (status == 0) & ((flags & 1) == 0)
    fi = np.nonzero(fast)[0]
    slow = [i for i in range(len(chunk)) if not fast[i]]
    res = dict(ci=ci, chunk=chunk, fi=fi, slow_records={})
    st = dict(res=res, be=be, slow=slow, h=h, t_start=t_start, eng=None)
    if len(fi):
        sizes = rows[fi] * cols[fi]
        raw_off = np.cumsum(sizes) - sizes
        t_off = np.cumsum(tbytes[fi]) - tbytes[fi]
        arena_buf, arena = be.pinned(int(sizes.sum()), ("arena", ci % DEPTH))
        titles = np.empty(max(int(tbytes[fi].sum()), 1), np.uint8)
        ro_all, to_all = np.full(len(chunk), -1, np.int64), np.zeros(len(chunk), np.int64)
        ro_all[fi], to_all[fi] = raw_off, t_off
        lib.mprg_ingest_fill_host(h, arena.ctypes.data, ro_all.ctypes.data, titles.ctypes.data, to_all.ctypes.data, threads)

        def ids_of(j):
            t = titles[t_off[j]:t_off[j] + tbytes[fi[j]]].tobytes().decode("ascii").split("\n")[:-1]
            return [(x.split(None, 1) or [""])[0] for x in t]

        t_fill = time.perf_counter()
        eng = ForestEngine(be, options.max_nesting, options.min_match_length)
        eng.load_raw(arena_buf, arena, raw_off, rows[fi], cols[fi], has_n=(flags[fi] & 2) != 0, ids_of=ids_of)
        eng.forest_enqueue()
        st.update(eng=eng, t_fill=t_fill, t_enq=time.perf_counter())
        res.update(eng=eng, arena=arena, raw_off=raw_off, rows=rows[fi], cols=cols[fi], titles=titles, t_off=t_off,
                   tbytes=tbytes[fi], site_count=None)
    return st


def _build_end(lib, options, threads, st, release):
    """Second half: waits for the chunk's forest, lays the PRG text and the tree export out and starts their copies to pinned
    memory; the files the native parser left alone take the object path here.  Returns what the output stage needs.
    release(h): gives the chunk's parser state back (off this thread: un-mapping ~300 MB takes 4-16 ms)."""
    from .subcommands import from_msa as drv
    import time
    res, be, ot = st["res"], st["be"], options.output_type
    ci, chunk = res["ci"], res["chunk"]
    t0 = time.perf_counter()
    if st["eng"] is not None:
        eng = st["eng"]
        eng.forest_finish()
        t_load = time.perf_counter()
        fin = eng.assemble_prgs(as_bytes=True, lazy=True, export=ot.prg, pack_alignments=ot.prg and PACK_ALIGNMENTS)
        _trace(f"chunk {ci}: fill {1e3 * (st['t_fill'] - st['t_start']):.0f} ms, load + enqueue {1e3 * (st['t_enq'] - st['t_fill']):.0f} ms, "
               f"forest done after {1e3 * (t_load - st['t_enq']):.0f} ms (waited {1e3 * (t_load - t0):.0f} ms), "
               f"assemble {1e3 * (time.perf_counter() - t_load):.0f} ms ({len(res['fi'])} alignments)")
        res.update(fin=fin)
    if st["slow"]:          # gzip / non-ASCII / duplicate ids / other formats: the object path, inside this chunk
        loaded = [drv._load_one((chunk[i], options.alignment_format)) for i in st["slow"]]
        msas, loci = [], []
        for i, m in zip(st["slow"], loaded):
            locus = remove_known_input_extensions(chunk[i].name)
            if isinstance(m, ValueError):
                if "No records found in handle" in str(m.args[0]):
                    raise drv.EmptyMSAError(f"No records found in MSA of locus {locus}")
                raise m
            msas.append(m)
            loci.append(locus)
        # (its copies cycle through pinned buffers of their OWN: the main ring holds the text / tree export of this chunk and of the
        #  two before it, which the output stage and the writers may still be reading)
        drv._build_batch(msas, loci, options, be, res["slow_records"], ring=1)
    if st["h"] is not None:
        release(st["h"])
    return (res,)


def _addr(a) -> int:
    return np.frombuffer(a, np.uint8).ctypes.data if len(a) else 0


def _write_chunk(lib, out: _Outputs, options, threads, res):
    """Output stage of one chunk: encoders + CRCs by native threads; the containers' members as ADDRESS TABLES over the chunk's
    buffers (array arithmetic for the loci of the arena path, a few Python bytes for the rare others).  Returns the write jobs
    (places already taken, in chunk order)."""
    import time
    import zlib
    tw0 = time.perf_counter()
    ot = options.output_type
    chunk, fi = res["chunk"], res["fi"]
    n_chunk, n_fast = len(chunk), len(fi)
    names = [remove_known_input_extensions(p.name) for p in chunk]
    keep = [res]                       # everything the tables point into
    # per locus of the chunk (in its sorted order): where its PRG text lies; -1 = no PRG
    t_addr, t_len = np.zeros(n_chunk, np.int64), np.full(n_chunk, -1, np.int64)
    tw1 = tw0
    if n_fast:
        eng, fin = res["eng"], res["fin"]
        prgs = fin()                                  # waits for the chunk's copies (text, tree export)
        tw1 = time.perf_counter()
        length = np.ascontiguousarray(fin.length, np.int64)
        base = np.ascontiguousarray(fin.base, np.int64)
        for j in np.nonzero(length < 0)[0].tolist():
            err = eng.errors[j]
            if not isinstance(err, SequenceCurationError):
                raise err
            logger.warning(f"Skipping building PRG for {names[int(fi[j])]}. Error: {err}")
        whole = np.frombuffer(fin.buffer, np.uint8) if len(fin.buffer) else np.zeros(1, np.uint8)
        keep.append(whole)
        # every locus encoded once, into the chunk's encode pool; members are written from the addresses
        bin_words, gfa_bytes = np.zeros(n_fast, np.int64), np.zeros(n_fast, np.int64)
        bin_addr, gfa_addr = np.zeros(n_fast, np.int64), np.zeros(n_fast, np.int64)
        crc = np.zeros((n_fast, 3), np.uint32)
        pool = out.pool(lib, res["ci"] % DEPTH)
        if lib.mprg_encode_batch_host(pool, whole.ctypes.data, base.ctypes.data, length.ctypes.data, n_fast, threads, int(ot.binary),
                                      int(ot.gfa), bin_addr.ctypes.data, bin_words.ctypes.data, gfa_addr.ctypes.data,
                                      gfa_bytes.ctypes.data, crc.ctypes.data) != 0:
            raise MemoryError("the encoders' pool could not grow")
        t_addr[fi], t_len[fi] = whole.ctypes.data + base, length
    tw2 = time.perf_counter()
    # ---- the loci of the object path and the PRGs the one-pass encoders do not cover: bytes, one by one (rare)
    extra = {"bin": [], "gfa": [], "pickle": []}          # (member name, bytes)
    fast_set = set(fi.tolist()) if n_fast < n_chunk else None
    for i in ([i for i in range(n_chunk) if i not in fast_set] if fast_set is not None else ()):
        rec = res["slow_records"].get(names[i])
        if rec is None:
            continue
        text = rec["prg"].encode()
        keep.append(text)
        t_addr[i], t_len[i] = _addr(text), len(text)
        for kind in extra:
            if kind in rec:
                extra[kind].append((i, names[i] if kind == "pickle" else f"{names[i]}.{kind}", rec[kind]))
    ok = np.zeros(0, bool)
    if n_fast:
        ok = length >= 0
        for j in np.nonzero(ok & (bin_words < 0) & bool(ot.binary))[0].tolist():      # the reference-shaped encoders own these strings
            from .utils.prg_encoder import PrgEncoder
            extra["bin"].append((int(fi[j]), names[int(fi[j])] + ".bin", np.asarray(PrgEncoder().encode(bytes(prgs[j]).decode()), "<u4").tobytes()))
        for j in np.nonzero(ok & (gfa_bytes < 0) & bool(ot.gfa))[0].tolist():
            from .utils.gfa import GFA_Output
            extra["gfa"].append((int(fi[j]), names[int(fi[j])] + ".gfa", GFA_Output.gfa_bytes(bytes(prgs[j]).decode())))
    built = np.nonzero(t_len >= 0)[0]
    out.n += len(built)
    jobs = []

    def timed(name, fn):
        def run(keep=keep):
            t0 = time.perf_counter()
            fn()
            _trace(f"chunk {res['ci']}: wrote {name} in {1e3 * (time.perf_counter() - t0):.0f} ms")
        return run

    def zip_job(kind, member_names, addr, ln, crcs, place):
        """place: the chunk position of every member's locus — the archive lists a chunk's members in the run's locus order."""
        place = list(place)
        for pos, nm, data in extra[kind]:     # the rare bytes members ride along: one more row each, at their locus's place
            keep.append(data)
            row_a, row_l = np.zeros((1, addr.shape[1]), np.int64), np.zeros((1, addr.shape[1]), np.int64)
            row_a[0, 0], row_l[0, 0] = _addr(data), len(data)
            addr, ln = np.concatenate([addr, row_a]), np.concatenate([ln, row_l])
            member_names = member_names + [nm]
            crcs = np.concatenate([crcs, [zlib.crc32(data)]])
            place.append(pos)
        if extra[kind]:
            order = np.argsort(np.asarray(place, np.int64), kind="stable")
            addr, ln, crcs = addr[order], ln[order], np.asarray(crcs)[order]
            member_names = [member_names[q] for q in order.tolist()]
        if len(member_names):
            jobs.append(timed(kind, out.zip(kind).plan_table(member_names, addr, ln, crcs, lib, WRITE_THREADS, keep)))

    # ---- <prefix>.prg.fa: ">locus\n" PRG "\n" per built locus, in order
    if ot.prg and len(built):
        heads = [(">" + names[i] + "\n").encode() for i in built.tolist()]
        blob = np.frombuffer(b"".join(heads) + b"\n", np.uint8)
        hl = np.fromiter((len(h) for h in heads), np.int64, len(heads))
        addr = np.empty((len(built), 3), np.int64)
        ln = np.empty_like(addr)
        addr[:, 0], ln[:, 0] = blob.ctypes.data + np.cumsum(hl) - hl, hl
        addr[:, 1], ln[:, 1] = t_addr[built], t_len[built]
        addr[:, 2], ln[:, 2] = blob.ctypes.data + blob.size - 1, 1
        keep.append(blob)
        out.fa_index.extend(zip((names[i] for i in built.tolist()), ln.sum(axis=1).tolist()))
        jobs.append(timed("prg.fa", out.plan_fa(addr.reshape(-1), ln.reshape(-1), lib, WRITE_THREADS)))
    # ---- zip members of the arena path's loci
    okj = np.nonzero(ok)[0]
    ok_names = [names[i] for i in fi[okj].tolist()] if n_fast else []
    if ot.binary:
        sel = okj[bin_words[okj] >= 0] if n_fast else okj
        zip_job("bin", [names[i] + ".bin" for i in fi[sel].tolist()] if n_fast else [],
                bin_addr[sel].reshape(-1, 1) if n_fast else np.zeros((0, 1), np.int64),
                (4 * bin_words[sel]).reshape(-1, 1) if n_fast else np.zeros((0, 1), np.int64), crc[sel, 1] if n_fast else np.zeros(0, np.uint32),
                fi[sel].tolist() if n_fast else [])
    if ot.gfa:
        sel = okj[gfa_bytes[okj] >= 0] if n_fast else okj
        zip_job("gfa", [names[i] + ".gfa" for i in fi[sel].tolist()] if n_fast else [],
                gfa_addr[sel].reshape(-1, 1) if n_fast else np.zeros((0, 1), np.int64),
                gfa_bytes[sel].reshape(-1, 1) if n_fast else np.zeros((0, 1), np.int64), crc[sel, 2] if n_fast else np.zeros(0, np.uint32),
                fi[sel].tolist() if n_fast else [])
    if ot.prg:          # update_DS members: header + slices of the arena, the titles and the device's tree export
        K = 7
        n_m = len(okj)
        addr, ln = np.zeros((n_m, K), np.int64), np.zeros((n_m, K), np.int64)
        crcs = np.zeros(n_m, np.uint32)
        if n_m:
            ex = eng.exported
            site = eng.site_count[okj]
            S, C, ro, to, tb = (res[k][okj] for k in ("rows", "cols", "raw_off", "t_off", "tbytes"))
            nb, rb, ib = ex["node_bounds"], ex["row_bounds"], ex["index_bounds"]
            n_nodes, n_rows, n_ix = nb[okj + 1] - nb[okj], rb[okj + 1] - rb[okj], ib[okj + 1] - ib[okj]
            extra_ix = [np.asarray(eng._host_index[j], np.int32).tobytes() if j in eng._host_index else b"" for j in okj.tolist()] \
                if eng._host_index else None
            fmt, N_, L_ = options.alignment_format, options.max_nesting, options.min_match_length
            packed = "alignments" in ex
            heads = [member_header(nm, fmt, N_, L_, a, 5 + 2 * b, c, d, e, a, f, g + (len(extra_ix[q]) // 12 if extra_ix else 0),
                                   enc="nib4" if packed else None)
                     for q, (nm, a, b, c, d, e, f, g) in enumerate(zip(ok_names, n_nodes.tolist(), site.tolist(), S.tolist(), C.tolist(),
                                                                        tb.tolist(), n_rows.tolist(), n_ix.tolist()))]
            blob = np.frombuffer(b"".join(heads), np.uint8)
            hl = np.fromiter((len(h) for h in heads), np.int64, n_m)
            keep += [blob, res["titles"]]
            addr[:, 0], ln[:, 0] = blob.ctypes.data + np.cumsum(hl) - hl, hl
            if packed:          # the alignment at four bits per cell, packed by the device (mprg_export_alignments)
                addr[:, 1], ln[:, 1] = ex["alignments"].ctypes.data + ex["alignment_off"][okj], ex["alignment_bytes"][okj]
            else:
                addr[:, 1], ln[:, 1] = res["arena"].ctypes.data + ro, S * C
            addr[:, 2], ln[:, 2] = res["titles"].ctypes.data + to, tb
            addr[:, 3], ln[:, 3] = ex["records"].ctypes.data + 32 * nb[okj], 32 * n_nodes
            addr[:, 4], ln[:, 4] = ex["rows"].ctypes.data + 4 * rb[okj], 4 * n_rows
            addr[:, 5], ln[:, 5] = ex["index"].ctypes.data + 12 * ib[okj], 12 * n_ix
            if extra_ix:
                for q, b_ in enumerate(extra_ix):
                    if b_:
                        keep.append(b_)
                        addr[q, 6], ln[q, 6] = _addr(b_), len(b_)
            first = np.arange(n_m + 1, dtype=np.int64) * K
            a_flat, l_flat = np.ascontiguousarray(addr.reshape(-1)), np.ascontiguousarray(ln.reshape(-1))
            lib.mprg_crc32_members_host(a_flat.ctypes.data, l_flat.ctypes.data, first.ctypes.data, n_m, threads, crcs.ctypes.data)
        zip_job("pickle", ok_names, addr, ln, crcs, fi[okj].tolist() if n_fast else [])
    _trace(f"chunk {res['ci']}: wait for copies {1e3 * (tw1 - tw0):.0f} ms, encode {1e3 * (tw2 - tw1):.0f} ms, "
           f"tables + places {1e3 * (time.perf_counter() - tw2):.0f} ms")
    return jobs

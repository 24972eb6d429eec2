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
CHUNK = int(os.environ.get("MPRG_CHUNK", "4096"))          # alignment files per resident batch
DEPTH = 3                                                  # chunks in flight: build | encode | write
TRACE = os.environ.get("MPRG_PIPELINE_TRACE", "") not in ("", "0")


def _trace(msg):
    if TRACE:
        import sys
        sys.stderr.write("[pipeline] " + msg + "\n")


def sort_key(path: Path) -> str:
    return remove_known_input_extensions(path.name) + ".prg.fa"          # the reference sorts its per-locus temp paths


class _Outputs:
    """The run's containers, streamed."""

    def __init__(self, prefix: str, ot, threads: int):
        self.prefix, self.ot = prefix, ot
        self.fa_fd = None
        self.zips = {}
        self.threads = threads
        self.n = 0
        self.last = None

    def zip(self, kind):
        if kind not in self.zips:
            name = f"{self.prefix}.update_DS.zip" if kind == "pickle" else f"{self.prefix}.prg.{kind}.zip"
            self.zips[kind] = StoredZipWriter(name, threads=max(2, min(8, self.threads)))
        return self.zips[kind]

    def plan_fa(self, pieces: List):
        """Places the pieces at the end of <prefix>.prg.fa and returns the function that writes them there."""
        if self.fa_fd is None:
            self.fa_fd = os.open(self.prefix + ".prg.fa", os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            self.fa_off = 0
        at, fd = self.fa_off, self.fa_fd
        self.fa_off += sum(len(p) for p in pieces)

        def run():
            off = at
            for p in pieces:
                mv = memoryview(p)
                done = 0
                while done < len(mv):
                    done += os.pwrite(fd, mv[done:], off + done)
                off += len(mv)

        return run

    def close(self):
        if self.fa_fd is not None:
            os.close(self.fa_fd)
        for z in self.zips.values():
            z.close()
        if self.n == 1:          # a single locus is written bare, not zipped (utils/input_output_files.py:104-131)
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


def run_pipeline(files: List[Path], options, backend) -> int:
    """Builds every locus of `files` and writes the run's output files.  Returns the number of loci built.
    backend: a backend object, or a function that makes one — it is called AFTER the ingest thread has started, so that reading
    and parsing the first chunks overlaps importing torch and bringing up the device (~2 s of a command-line run)."""
    from .subcommands import from_msa as drv
    lib = native.library()
    if lib is None:
        raise RuntimeError("libmprg_host.so is missing: build it with `python __graft_entry__.py`")
    ot = options.output_type
    threads = max(1, int(getattr(options, "threads", 1) or 1))
    files = sorted(files, key=sort_key)
    chunks = [files[lo:lo + CHUNK] for lo in range(0, len(files), CHUNK)]
    out = _Outputs(options.output_prefix, ot, threads)
    fasta = options.alignment_format == "fasta"
    q_in: "queue.Queue" = queue.Queue(maxsize=2)
    q_out: "queue.Queue" = queue.Queue()
    errors: List[BaseException] = []
    # a chunk's pinned buffers (arena, PRG text, tree export: DEPTH of each, used in turn) are read by the output stages until
    # its members are written: the build of chunk i + DEPTH starts only when the writes of chunk i are done
    in_flight = threading.Semaphore(DEPTH)
    writers = ThreadPoolExecutor(4)

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
                jobs = _write_chunk(lib, out, options, threads, *item)
                # the containers are written side by side while the next chunk is encoded; the chunk's buffers are free again
                # once all of them are done
                futs = [writers.submit(j) for j in jobs]
                pending = [len(futs)]
                lock = threading.Lock()

                def done(f, pending=pending, lock=lock):
                    if f.exception() is not None:
                        errors.append(f.exception())
                    with lock:
                        pending[0] -= 1
                        last = pending[0] == 0
                    if last:
                        in_flight.release()

                if not futs:
                    in_flight.release()
                for f in futs:
                    f.add_done_callback(done)
        except BaseException as err:
            errors.append(err)
            in_flight.release()
            while q_out.get() is not None:
                in_flight.release()

    t_in, t_out = threading.Thread(target=stage_ingest, daemon=True), threading.Thread(target=stage_output, daemon=True)
    t_in.start()
    t_out.start()
    be = backend() if callable(backend) else backend
    be.async_depth = DEPTH
    try:
        while True:
            item = q_in.get()
            if item is None or errors:
                break
            ci, chunk, h, info = item
            in_flight.acquire()
            if errors:
                break
            q_out.put(_build_chunk(lib, be, options, threads, ci, chunk, h, info))
    finally:
        q_out.put(None)
        t_out.join()
        writers.shutdown(wait=True)
    if errors:
        raise errors[0]
    out.close()
    return out.n


def _build_chunk(lib, be, options, threads, ci, chunk, h, info):
    """Device stage of one chunk: fast files through the arena, the rest through the object path.  Returns what the output
    stage needs."""
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
        eng.run_forest()
        t_load = time.perf_counter()
        fin = eng.assemble_prgs(as_bytes=True, lazy=True, export=ot.prg)
        _trace(f"chunk {ci}: fill {1e3 * (t_fill - t_start):.0f} ms, load + forest {1e3 * (t_load - t_fill):.0f} ms, "
               f"assemble {1e3 * (time.perf_counter() - t_load):.0f} ms ({len(fi)} alignments)")
        res.update(eng=eng, fin=fin, arena=arena, raw_off=raw_off, rows=rows[fi], cols=cols[fi], titles=titles, t_off=t_off,
                   tbytes=tbytes[fi], site_count=None)
    if slow:          # gzip / non-ASCII / duplicate ids / other formats: the object path, inside this chunk
        loaded = [drv._load_one((chunk[i], options.alignment_format)) for i in slow]
        msas, loci = [], []
        for i, m in zip(slow, loaded):
            locus = remove_known_input_extensions(chunk[i].name)
            if isinstance(m, ValueError):
                if "No records found in handle" in str(m.args[0]):
                    raise drv.EmptyMSAError(f"No records found in MSA of locus {locus}")
                raise m
            msas.append(m)
            loci.append(locus)
        drv._build_batch(msas, loci, options, be, res["slow_records"])
    if h is not None:
        lib.mprg_ingest_close_host(h)
    return (res,)


def _write_chunk(lib, out: _Outputs, options, threads, res):
    """Output stage of one chunk: encoders + CRCs by native threads, members streamed into the containers."""
    import time
    tw0 = time.perf_counter()
    ot = options.output_type
    chunk, fi = res["chunk"], res["fi"]
    names = [remove_known_input_extensions(p.name) for p in chunk]
    n_fast = len(fi)
    fast_pos = {int(i): j for j, i in enumerate(fi.tolist())}
    prgs = []
    buf0_ok = False
    if n_fast:
        eng = res["eng"]
        prgs = res["fin"]()                           # waits for the chunk's copies; memoryviews into the pinned text buffer
        for j in range(n_fast):
            if prgs[j] is None:
                err = eng.errors[j]
                if not isinstance(err, SequenceCurationError):
                    raise err
                logger.warning(f"Skipping building PRG for {names[int(fi[j])]}. Error: {err}")
        tw1 = time.perf_counter()
        fin = res["fin"]
        whole = np.frombuffer(fin.buffer, np.uint8) if len(fin.buffer) else np.zeros(1, np.uint8)
        base, length = np.ascontiguousarray(fin.base, np.int64), np.ascontiguousarray(fin.length, np.int64)
        buf0_ok = True
        if True:
            bin_words, gfa_bytes = np.zeros(n_fast, np.int64), np.zeros(n_fast, np.int64)
            lib.mprg_encode_sizes_host(whole.ctypes.data, base.ctypes.data, length.ctypes.data, n_fast, threads, int(ot.binary),
                                       int(ot.gfa), bin_words.ctypes.data, gfa_bytes.ctypes.data)
            bw, gb = np.maximum(bin_words, 0), np.maximum(gfa_bytes, 0)
            bin_off, gfa_off = np.cumsum(bw) - bw, np.cumsum(gb) - gb
            bin_buf = np.empty(max(int(bw.sum()), 1), np.uint32) if ot.binary else None
            gfa_buf = np.empty(max(int(gb.sum()), 1), np.uint8) if ot.gfa else None
            crc = np.zeros((n_fast, 3), np.uint32)
            lib.mprg_encode_fill_host(whole.ctypes.data, base.ctypes.data, length.ctypes.data, n_fast, threads,
                                      bin_buf.ctypes.data if ot.binary else None, bin_off.ctypes.data, bin_words.ctypes.data,
                                      gfa_buf.ctypes.data if ot.gfa else None, gfa_off.ctypes.data, gfa_bytes.ctypes.data, crc.ctypes.data)
    tw2 = time.perf_counter()
    # ---- members in the chunk's (sorted) locus order; the rare loci of the object path and those the one-pass encoders do not
    #      cover come as bytes
    fa, zb, zg, zp = [], ([], [], [], []), ([], [], [], []), ([], [], [], [])
    pk_jobs = []
    # (plain Python ints and one memoryview per big buffer: NumPy scalars and array slices would dominate this loop)
    if n_fast and buf0_ok:
        bw_l, bo_l, gb_l, go_l = bin_words.tolist(), bin_off.tolist(), gfa_bytes.tolist(), gfa_off.tolist()
        crc_l = crc.tolist()
        bin_mv = memoryview(bin_buf).cast("B") if ot.binary else None
        gfa_mv = memoryview(gfa_buf) if ot.gfa else None
    for i, locus in enumerate(names):
        j = fast_pos.get(i)
        if j is None:
            rec = res["slow_records"].get(locus)
            if rec is None:
                continue
            out.n += 1
            text = rec["prg"].encode()
            fa += [(">" + locus + "\n").encode(), text, b"\n"]
            for kind, dst in (("bin", zb), ("gfa", zg), ("pickle", zp)):
                if kind in rec:
                    import zlib
                    dst[0].append(locus if kind == "pickle" else f"{locus}.{kind}")
                    dst[1].append([rec[kind]]); dst[3].append(len(rec[kind]))
                    if kind != "pickle":          # (the update_DS members' CRCs are computed together below)
                        dst[2].append(zlib.crc32(rec[kind]))
            continue
        p = prgs[j]
        if p is None:
            continue
        out.n += 1
        if ot.prg:
            fa += [(">" + locus + "\n").encode(), p, b"\n"]
        if ot.binary:
            if bw_l[j] >= 0:
                piece = bin_mv[4 * bo_l[j]:4 * (bo_l[j] + bw_l[j])]
                c = crc_l[j][1]
            else:          # the reference-shaped encoder owns this string (and its errors)
                from .utils.prg_encoder import PrgEncoder
                import zlib
                piece = np.asarray(PrgEncoder().encode(bytes(p).decode()), "<u4").tobytes()
                c = zlib.crc32(piece)
            zb[0].append(locus + ".bin"); zb[1].append([piece]); zb[2].append(c); zb[3].append(len(piece))
        if ot.gfa:
            if gb_l[j] >= 0:
                piece = gfa_mv[go_l[j]:go_l[j] + gb_l[j]]
                c = crc_l[j][2]
            else:
                from .utils.gfa import GFA_Output
                import zlib
                piece = GFA_Output.gfa_bytes(bytes(p).decode())
                c = zlib.crc32(piece)
            zg[0].append(locus + ".gfa"); zg[1].append([piece]); zg[2].append(c); zg[3].append(len(piece))
        if ot.prg:
            pk_jobs.append((locus, j))
    if pk_jobs:          # update_DS members: header + slices of the arena, the titles and the device's tree export
        eng, ex = res["eng"], res["eng"].exported
        site_l = eng.site_count.tolist()
        arena, titles = res["arena"], res["titles"]
        ro_l, S_l, C_l, to_l, tb_l = (res[k].tolist() for k in ("raw_off", "rows", "cols", "t_off", "tbytes"))
        nb, rb, ib = ex["node_bounds"].tolist(), ex["row_bounds"].tolist(), ex["index_bounds"].tolist()
        recs_b = memoryview(ex["records"]).cast("B")
        rows_b = memoryview(ex["rows"]).cast("B") if len(ex["rows"]) else memoryview(b"")
        index_b = memoryview(ex["index"]).cast("B") if len(ex["index"]) else memoryview(b"")
        arena_b, titles_b = memoryview(arena), memoryview(titles)
        fmt, N_, L_ = options.alignment_format, options.max_nesting, options.min_match_length
        for locus, j in pk_jobs:
            n_nodes, n_rows, n_ix = nb[j + 1] - nb[j], rb[j + 1] - rb[j], ib[j + 1] - ib[j]
            extra = eng._host_index.get(j)
            S, C = S_l[j], C_l[j]
            ix_extra = np.asarray(extra, np.int32).tobytes() if extra else b""
            head = member_header(locus, fmt, N_, L_, n_nodes, 5 + 2 * site_l[j], S, C, tb_l[j], n_nodes, n_rows, n_ix + len(ix_extra) // 12)
            pieces = [head, arena_b[ro_l[j]:ro_l[j] + S * C], titles_b[to_l[j]:to_l[j] + tb_l[j]],
                      recs_b[32 * nb[j]:32 * nb[j + 1]], rows_b[4 * rb[j]:4 * rb[j + 1]], index_b[12 * ib[j]:12 * ib[j + 1]]]
            size = len(head) + S * C + tb_l[j] + 32 * n_nodes + 4 * n_rows + 12 * n_ix
            if ix_extra:
                pieces.append(ix_extra)
                size += len(ix_extra)
            zp[0].append(locus); zp[1].append(pieces); zp[3].append(size)
    if zp[0]:            # CRC-32 of the update_DS members by the native threads: one running value over a member's pieces
        zp[2][:] = _crc_members(lib, zp[1], threads)
    tw3 = time.perf_counter()
    _trace(f"chunk {res['ci']}: wait for text {1e3 * ((tw1 if n_fast else tw0) - tw0):.0f} ms, encode {1e3 * (tw2 - (tw1 if n_fast else tw0)):.0f} ms, "
           f"members {1e3 * (tw3 - tw2):.0f} ms")
    jobs = []
    keep = res                      # the chunk's buffers stay referenced by the jobs until they have run

    def timed(name, fn):
        def run(keep=keep):
            t0 = time.perf_counter()
            fn()
            _trace(f"chunk {res['ci']}: wrote {name} in {1e3 * (time.perf_counter() - t0):.0f} ms")
        return run

    # places are taken here, in chunk order; the copies run later, side by side with those of the neighbouring chunks
    if fa and ot.prg:
        jobs.append(timed("prg.fa", out.plan_fa(fa)))
    for kind, dst, want in (("bin", zb, ot.binary), ("gfa", zg, ot.gfa), ("pickle", zp, ot.prg)):
        if want and dst[0]:
            jobs.append(timed(kind, out.zip(kind).plan_many(*dst)))
    return jobs


def _crc_members(lib, members: List[List], threads: int) -> List[int]:
    """CRC-32 of members given as lists of pieces, by the native threads (one running value per member)."""
    n = sum(len(pieces) for pieces in members)
    addr, ln, first = np.empty(n, np.int64), np.empty(n, np.int64), np.zeros(len(members) + 1, np.int64)
    keep, k = [], 0
    for i, pieces in enumerate(members):
        for p in pieces:
            if len(p):
                a = np.frombuffer(p, np.uint8)
                keep.append(a)
                addr[k], ln[k] = a.ctypes.data, a.size
            else:
                addr[k], ln[k] = 0, 0
            k += 1
        first[i + 1] = k
    crc = np.zeros(len(members), np.uint32)
    lib.mprg_crc32_members_host(addr.ctypes.data, ln.ctypes.data, first.ctypes.data, len(members), threads, crc.ctypes.data)
    return [int(c) for c in crc]

"""Streaming STORED zip writer of the from_msa driver's containers (reference utils/io_utils.py:105-110 zips a directory of
per-locus temp files at the end of the run; utils/input_output_files.py:73-162).

Members arrive batch by batch as lists of buffers (slices of the batch's big arrays) with their CRC-32 already computed by the
native encoders' threads, are placed by a running offset and written with positional writes from a few threads — a stored
member is a header and a copy, nothing is concatenated or compressed, and the file is never held in memory.  ZIP64 records are
written when the archive outgrows 4 GiB or 65 535 members (a 30 000-locus .prg.bin.zip is ~10 GB).  Timestamps are fixed
(1980-01-01) so that equal inputs give byte-identical archives."""
import os
import struct
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence

_LOCAL, _CENTRAL, _EOCD, _EOCD64, _LOC64 = 0x04034B50, 0x02014B50, 0x06054B50, 0x06064B50, 0x07064B50
_DOSDATE = (1980 - 1980) << 9 | 1 << 5 | 1


class StoredZipWriter:
    def __init__(self, path, threads: int = 4):
        self.path = str(path)
        self.fd = None
        self.offset = 0
        self.entries = []           # (name bytes, crc, size, offset)
        self.pool = ThreadPoolExecutor(max(1, threads))

    def _open(self):
        if self.fd is None:
            self.fd = os.open(self.path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)

    @staticmethod
    def _local_header(name: bytes, crc: int, size: int) -> bytes:
        return struct.pack("<IHHHHHIIIHH", _LOCAL, 20, 0, 0, 0, _DOSDATE, crc & 0xFFFFFFFF, size, size, len(name), 0) + name

    def plan_many(self, names: Sequence[str], buffers: Sequence[List], crcs: Sequence[int], sizes: Sequence[int]):
        """Places members (the cheap, ordered part: offsets, directory entries) and returns the function that writes them —
        callers may run the writers of successive batches concurrently: the places are disjoint.
        buffers[i] = the list of bytes-like pieces of member i (their total length is sizes[i] < 4 GiB)."""
        self._open()
        jobs = []
        for name, bufs, crc, size in zip(names, buffers, crcs, sizes):
            nb = name.encode("utf-8")
            head = self._local_header(nb, int(crc), int(size))
            jobs.append((self.offset, [head] + list(bufs)))
            self.entries.append((nb, int(crc) & 0xFFFFFFFF, int(size), self.offset))
            self.offset += len(head) + int(size)
        fd = self.fd

        def run():
            if not jobs:
                return
            n_part = min(len(jobs), self.pool._max_workers * 4)
            step = (len(jobs) + n_part - 1) // n_part

            def write(lo):
                for off, pieces in jobs[lo:lo + step]:
                    for p in pieces:                      # (os.pwritev would need the pieces' total under 2 GiB and IOV_MAX)
                        mv = memoryview(p)
                        done = 0
                        while done < len(mv):
                            done += os.pwrite(fd, mv[done:], off + done)
                        off += len(mv)

            list(self.pool.map(write, range(0, len(jobs), step)))

        return run

    def add_many(self, names, buffers, crcs, sizes):
        self.plan_many(names, buffers, crcs, sizes)()

    def plan_table(self, names: Sequence[str], data_addr, data_len, crcs, lib, threads: int, keep=None):
        """plan_many for members given as ADDRESS TABLES (int64 [members, pieces]: where each piece of a member's data lies in
        memory and how long it is) — the batch's members are slices of a few big buffers, so the tables are array arithmetic and
        the copies one call of libmprg's scatter-gather writer (mprg_write_pieces_host, `threads` native threads, no GIL).
        keep: whatever must stay alive until the returned function has run."""
        import numpy as np
        self._open()
        n = len(names)
        data_addr = np.ascontiguousarray(data_addr, np.int64).reshape(n, -1)
        data_len = np.ascontiguousarray(data_len, np.int64).reshape(n, -1)
        sizes = data_len.sum(axis=1)
        heads = [self._local_header(nm.encode("utf-8"), int(c), int(sz)) for nm, c, sz in zip(names, np.asarray(crcs).tolist(), sizes.tolist())]
        blob = np.frombuffer(b"".join(heads), np.uint8) if n else np.zeros(0, np.uint8)
        hl = np.fromiter((len(h) for h in heads), np.int64, n)
        addr = np.empty((n, data_addr.shape[1] + 1), np.int64)
        ln = np.empty_like(addr)
        addr[:, 0] = blob.ctypes.data + np.cumsum(hl) - hl
        ln[:, 0] = hl
        addr[:, 1:], ln[:, 1:] = data_addr, data_len
        flat_len = ln.reshape(-1)
        off = self.offset + np.cumsum(flat_len) - flat_len
        starts = off.reshape(n, -1)[:, 0].tolist() if n else []
        for nm, c, sz, st in zip(names, np.asarray(crcs).tolist(), sizes.tolist(), starts):
            self.entries.append((nm.encode("utf-8"), int(c) & 0xFFFFFFFF, int(sz), int(st)))
        self.offset += int(flat_len.sum())
        fd = self.fd
        addr, off = np.ascontiguousarray(addr.reshape(-1)), np.ascontiguousarray(off)

        def run(keep=(keep, blob)):
            if n and lib.mprg_write_pieces_host(fd, addr.ctypes.data, flat_len.ctypes.data, off.ctypes.data, len(flat_len), threads) != 0:
                raise OSError(f"writing {self.path} failed")

        return run

    def add(self, name: str, data: bytes, crc: int = None):
        import zlib
        self.add_many([name], [[data]], [zlib.crc32(data) if crc is None else crc], [len(data)])

    def close(self):
        self.pool.shutdown(wait=True)
        if self.fd is None:
            return
        cd = bytearray()
        for nb, crc, size, off in self.entries:
            extra = b""
            off32 = off
            if off >= 0xFFFFFFFF:
                extra = struct.pack("<HHQ", 1, 8, off)
                off32 = 0xFFFFFFFF
            cd += struct.pack("<IHHHHHHIIIHHHHHII", _CENTRAL, 45 if extra else 20, 45 if extra else 20, 0, 0, 0, _DOSDATE, crc, size, size,
                              len(nb), len(extra), 0, 0, 0, 0o600 << 16, off32)
            cd += nb + extra
        cd_off, n = self.offset, len(self.entries)
        tail = bytearray()
        if cd_off >= 0xFFFFFFFF or len(cd) >= 0xFFFFFFFF or n >= 0xFFFF:
            tail += struct.pack("<IQHHIIQQQQ", _EOCD64, 44, 45, 45, 0, 0, n, n, len(cd), cd_off)
            tail += struct.pack("<IIQI", _LOC64, 0, cd_off + len(cd), 1)
        tail += struct.pack("<IHHHHIIH", _EOCD, 0, 0, min(n, 0xFFFF), min(n, 0xFFFF), min(len(cd), 0xFFFFFFFF), min(cd_off, 0xFFFFFFFF), 0)
        blob = bytes(cd) + bytes(tail)
        done = 0
        while done < len(blob):
            done += os.pwrite(self.fd, blob[done:], cd_off + done)
        os.close(self.fd)
        self.fd = None

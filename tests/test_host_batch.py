"""libmprg_host.so's batch stages on their own (no GPU): the folding CRC-32 against zlib, the pooled one-pass encoders against
the per-locus encoders, pool reuse."""
import os
import random
import zlib

import numpy as np
import pytest

from make_prg_amd.utils import native


@pytest.fixture(scope="module")
def lib():
    lib = native.library()
    if lib is None:
        pytest.skip("libmprg_host.so not built")
    return lib


def test_crc32_equals_zlib(lib):
    rng = np.random.default_rng(7)
    for n in list(range(0, 300)) + [1023, 4096, 65543, (1 << 20) + 5]:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for start in (0, 0x12345678):
            assert lib.mprg_crc32_host(start, data, n) == zlib.crc32(data, start), (n, start)
    # a running CRC over pieces of awkward sizes
    data = rng.integers(0, 256, 100000, dtype=np.uint8).tobytes()
    c, at = 0, 0
    for piece in (1, 63, 64, 65, 15, 16, 17, 1000, 33333):
        c = lib.mprg_crc32_host(c, data[at:at + piece], piece)
        at += piece
    assert c == zlib.crc32(data[:at])


def _random_prg(rnd, depth=0, site=[5]):
    def dna(lo, hi):
        return "".join(rnd.choice("ACGT") for _ in range(rnd.randint(lo, hi)))
    out = [dna(1, 40)]
    for _ in range(rnd.randint(1, 4)):
        s = site[0]
        site[0] += 2
        alleles = []
        for _ in range(rnd.randint(2, 4)):
            alleles.append(_random_prg(rnd, depth + 1, site) if depth < 2 and rnd.random() < 0.3 else dna(0 if depth else 1, 30))
        out.append(f" {s} " + f" {s + 1} ".join(alleles) + f" {s} " + dna(1, 40))
    return "".join(out)


def test_encode_batch_equals_per_locus(lib):
    rnd = random.Random(3)
    prgs = []
    for i in range(300):
        prgs.append(_random_prg(rnd, 0, [5]).encode())
    prgs[17] = b"ACGT 5 A 6 C 5 N"                 # not plain for the binary encoder (N), GFA takes it
    prgs[40] = b"AC 7 A 8 C 7 G"                   # sites must start at 5: not plain for the GFA form
    prgs[99] = ("A" * (40 << 20)).encode()         # one locus bigger than a pool block
    text = np.frombuffer(b"".join(prgs), np.uint8)
    length = np.array([len(p) for p in prgs], np.int64)
    base = np.cumsum(length) - length
    length[5] = -1                                  # no PRG for this locus
    n = len(prgs)
    pool = lib.mprg_encode_pool_new_host()
    try:
        for round_ in range(2):                     # the second round reuses the blocks
            if round_:
                lib.mprg_encode_pool_reset_host(pool)
            ba, bw, ga, gb = (np.zeros(n, np.int64) for _ in range(4))
            crc = np.zeros((n, 3), np.uint32)
            assert lib.mprg_encode_batch_host(pool, text.ctypes.data, base.ctypes.data, length.ctypes.data, n, 5, 1, 1, ba.ctypes.data,
                                              bw.ctypes.data, ga.ctypes.data, gb.ctypes.data, crc.ctypes.data) == 0
            import ctypes
            for i, p in enumerate(prgs):
                if length[i] < 0:
                    assert bw[i] == -1 and gb[i] == -1
                    continue
                want_bin = native.prg_encode(p)
                want_gfa = native.gfa_text(p)
                assert crc[i, 0] == zlib.crc32(p)
                if want_bin is None:
                    assert bw[i] == -1
                else:
                    got = ctypes.string_at(int(ba[i]), int(4 * bw[i]))
                    assert got == np.asarray(want_bin, "<u4").tobytes()
                    assert crc[i, 1] == zlib.crc32(got)
                if want_gfa is None:
                    assert gb[i] == -1
                else:
                    got = ctypes.string_at(int(ga[i]), int(gb[i]))
                    assert got == (want_gfa if isinstance(want_gfa, bytes) else want_gfa.encode())
                    assert crc[i, 2] == zlib.crc32(got)
            info = np.zeros(2, np.int64)
            lib.mprg_encode_pool_info_host(pool, info.ctypes.data)
            if round_ == 0:
                mapped = int(info[0])
            else:
                assert int(info[0]) == mapped       # nothing new was mapped for the same work
    finally:
        lib.mprg_encode_pool_free_host(pool)


def test_ingest_from_memory_equals_ingest_from_files(lib, tmp_path):
    """mprg_ingest_open_mem_host (texts in memory, nothing copied) against mprg_ingest_open_host on the same texts as files:
    same status / rows / columns / title bytes / flags, same matrices and titles."""
    import ctypes
    texts = [b">a x\nACGT-N\n>b\nacgtta\n", b">r1\nAC\nGT\n>r2\nACGT\n", b"\x1f\x8bnot really gzip", b">a\nAC\n>a\nAG\n", b"", b">x\nACG\n>y\nAC\n"]
    paths = []
    for i, t in enumerate(texts):
        p = tmp_path / f"f{i}.fa"
        p.write_bytes(t)
        paths.append(p)
    n = len(texts)
    blob = b"".join(os.fsencode(str(p)) + b"\0" for p in paths)
    h_file = lib.mprg_ingest_open_host(blob, n, 3)
    ptrs = (ctypes.c_char_p * n)(*texts)
    lens = np.array([len(t) for t in texts], np.int64)
    h_mem = lib.mprg_ingest_open_mem_host(ptrs, lens.ctypes.data, n, 3)
    info_f, info_m = np.zeros((n, 5), np.int64), np.zeros((n, 5), np.int64)
    lib.mprg_ingest_info_host(h_file, info_f.ctypes.data)
    lib.mprg_ingest_info_host(h_mem, info_m.ctypes.data)
    assert np.array_equal(info_f, info_m)
    assert info_f[:, 0].tolist() == [0, 0, -3, 0, -7, -5] and info_f[3, 4] & 1 and info_f[0, 4] & 2
    ok = info_f[:, 0] == 0
    sizes = np.where(ok, info_f[:, 1] * info_f[:, 2], 0)
    raw_off = np.where(ok, np.cumsum(sizes) - sizes, -1)
    t_off = np.cumsum(np.where(ok, info_f[:, 3], 0)) - np.where(ok, info_f[:, 3], 0)
    outs = []
    for h in (h_file, h_mem):
        arena, titles = np.zeros(int(sizes.sum()) + 1, np.uint8), np.zeros(int(info_f[ok, 3].sum()) + 1, np.uint8)
        lib.mprg_ingest_fill_host(h, arena.ctypes.data, raw_off.ctypes.data, titles.ctypes.data, t_off.ctypes.data, 2)
        outs.append((arena.tobytes(), titles.tobytes()))
        lib.mprg_ingest_close_host(h)
    assert outs[0] == outs[1] and outs[0][0].startswith(b"ACGT-NACGTTA") and outs[0][1].startswith(b"a x\nb\n")


def test_gzipped_files_are_inflated_by_the_native_parser(lib, tmp_path):
    """utils/io_utils.py:24-26: a path ending in .gz is read through gzip.  The native batch parser inflates such files itself (one
    or several gzip members) and hands them on like plain text; a damaged .gz is left to the Python parser (status -3)."""
    import gzip
    texts = [b">a x\nACGT-N\n>b\nacgtta\n", b">r1\n" + b"ACGT" * 5000 + b"\n>r2\n" + b"ACGA" * 5000 + b"\n"]
    paths = []
    for i, t in enumerate(texts):
        p = tmp_path / f"g{i}.fa.gz"
        p.write_bytes(gzip.compress(t))
        paths.append(p)
    p = tmp_path / "two_members.fa.gz"
    p.write_bytes(gzip.compress(b">m1\nACGT\n") + gzip.compress(b">m2\nACGA\n"))
    paths.append(p)
    p = tmp_path / "broken.fa.gz"
    p.write_bytes(gzip.compress(texts[0])[:-9])
    paths.append(p)
    p = tmp_path / "plain_magic.fa"          # gzip bytes without the .gz name: not inflated (the reference decides by name)
    p.write_bytes(gzip.compress(texts[0]))
    paths.append(p)
    n = len(paths)
    blob = b"".join(os.fsencode(str(q)) + b"\0" for q in paths)
    h = lib.mprg_ingest_open_host(blob, n, 2)
    info = np.zeros((n, 5), np.int64)
    lib.mprg_ingest_info_host(h, info.ctypes.data)
    assert info[:, 0].tolist() == [0, 0, 0, -3, -3]
    assert info[:3, 1].tolist() == [2, 2, 2] and info[:3, 2].tolist() == [6, 20000, 4] and info[0, 4] & 2
    ok = info[:, 0] == 0
    sizes = np.where(ok, info[:, 1] * info[:, 2], 0)
    raw_off = np.where(ok, np.cumsum(sizes) - sizes, -1)
    t_off = np.cumsum(np.where(ok, info[:, 3], 0)) - np.where(ok, info[:, 3], 0)
    arena, titles = np.zeros(int(sizes.sum()) + 1, np.uint8), np.zeros(int(info[ok, 3].sum()) + 1, np.uint8)
    lib.mprg_ingest_fill_host(h, arena.ctypes.data, raw_off.ctypes.data, titles.ctypes.data, t_off.ctypes.data, 2)
    lib.mprg_ingest_close_host(h)
    assert arena.tobytes().startswith(b"ACGT-NACGTTA" + b"ACGT" * 5000 + b"ACGA" * 5000 + b"ACGTACGA")
    assert titles.tobytes().startswith(b"a x\nb\nr1\nr2\nm1\nm2\n")

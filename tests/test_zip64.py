"""utils/zip_stream.py writes ZIP64 records when an archive outgrows 65 535 members or 4 GiB (a 30 000-locus .prg.bin.zip is ~10 GB):
both branches read back with the standard library's zipfile."""
import os
import zipfile
import zlib

from make_prg_amd.utils.zip_stream import StoredZipWriter


def test_more_than_65535_members(tmp_path):
    path = tmp_path / "many.zip"
    w = StoredZipWriter(path)
    n = 70_000
    names = [f"locus{i:06d}.bin" for i in range(n)]
    datas = [(b"%d" % i) * (1 + i % 3) for i in range(n)]
    w.add_many(names, [[d] for d in datas], [zlib.crc32(d) for d in datas], [len(d) for d in datas])
    w.close()
    with zipfile.ZipFile(path) as z:
        assert len(z.namelist()) == n and z.namelist()[0] == names[0] and z.namelist()[-1] == names[-1]
        for i in (0, 1, 65_534, 65_535, 65_536, n - 1):
            assert z.read(names[i]) == datas[i]
        assert z.testzip() is None


def test_member_beyond_four_gib(tmp_path):
    """Members whose local headers lie beyond 4 GiB: the archive starts with 4.2 GiB that are never written (a hole in a sparse file:
    unused space before the first member, which the format allows), so offsets and the central directory's place need ZIP64."""
    path = tmp_path / "big.zip"
    w = StoredZipWriter(path)
    w._open()
    hole = (4 << 30) + (200 << 20)
    w.offset = hole
    small = [b"ACGT" * 10, b"\x01\x00\x00\x00" * 7]
    w.add_many(["a.bin", "b.bin"], [[s] for s in small], [zlib.crc32(s) for s in small], [len(s) for s in small])
    w.close()
    assert os.path.getsize(path) > 4 << 30
    with zipfile.ZipFile(path) as z:
        assert z.namelist() == ["a.bin", "b.bin"]
        assert z.getinfo("a.bin").header_offset == hole and z.getinfo("a.bin").header_offset > 0xFFFFFFFF
        assert z.read("a.bin") == small[0] and z.read("b.bin") == small[1]
        assert z.testzip() is None

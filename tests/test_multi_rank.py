"""N > 1 path: ranks (gloo, CPU, emulation backend) shard a directory of alignments, every rank streams its shard into segment
files (the one-GPU pipeline), rank 0 gathers the segments' index and merges their byte ranges; every output file must equal the
single-process run's byte for byte."""
import os
import subprocess
import sys
import zipfile

from make_prg_amd.utils.synthetic import synth_config_fasta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_equal_one(tmp_path, golden_integration):
    d = tmp_path / "msas"
    d.mkdir()
    case = next(c for c in golden_integration["cases"] if c["case"] == "several")
    for l in case["loci"]:
        (d / l["file"]).write_text(l["fasta"])
    for s in range(5):
        (d / f"synth{s}.fa").write_text(synth_config_fasta("B", s))
    env = dict(os.environ, MPRG_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    one = tmp_path / "one" / "out"
    two = tmp_path / "two" / "out"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(d), str(one)], env=env)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533",
                           os.path.join(ROOT, "tests", "dist_worker.py"), str(d), str(two)], env=env)
    assert (tmp_path / "one" / "out.prg.fa").read_bytes() == (tmp_path / "two" / "out.prg.fa").read_bytes()
    for kind in ("prg.bin", "prg.gfa", "update_DS"):
        # the ranks' segments merged by rank 0 ARE the single-rank files: same member order, headers, CRCs, central directory
        assert open(f"{one}.{kind}.zip", "rb").read() == open(f"{two}.{kind}.zip", "rb").read(), kind
        with zipfile.ZipFile(f"{two}.{kind}.zip") as b:
            assert b.testzip() is None and len(b.namelist()) == 9
    assert sorted(p.name for p in (tmp_path / "two").iterdir()) == ["out.prg.bin.zip", "out.prg.fa", "out.prg.gfa.zip", "out.update_DS.zip"]


def test_four_ranks_with_skewed_file_sizes(tmp_path, golden_integration):
    """Four ranks (gloo), 26 alignments whose sizes span 40x: the shards must be balanced by size (longest-processing-time
    greedy: within 10 % of each other here) and the gathered result equal to the single-process run."""
    from pathlib import Path
    from make_prg_amd.subcommands.from_msa import balanced_parts
    d = tmp_path / "msas"
    d.mkdir()
    case = next(c for c in golden_integration["cases"] if c["case"] == "several")
    for l in case["loci"]:
        (d / l["file"]).write_text(l["fasta"])
    from make_prg_amd.utils.synthetic import synth_fasta
    for s in range(22):
        S, C = (12 + 3 * (s % 5), 60 + 25 * (s % 7)) if s % 4 else (40, 400)
        (d / f"skew{s}.fa").write_text(synth_fasta(s, S, C, 2 + s % 3))
    files = sorted(Path(d).iterdir())
    loads = [sum(f.stat().st_size for f in part) for part in balanced_parts(files, 4)]
    assert (max(loads) - min(loads)) / max(loads) <= 0.10, loads
    env = dict(os.environ, MPRG_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    one, four = tmp_path / "one" / "out", tmp_path / "four" / "out"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(d), str(one)], env=env)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
                           "--master-addr", "127.0.0.1", "--master-port", "29537",
                           os.path.join(ROOT, "tests", "dist_worker.py"), str(d), str(four)], env=env)
    assert (tmp_path / "one" / "out.prg.fa").read_bytes() == (tmp_path / "four" / "out.prg.fa").read_bytes()
    for kind in ("prg.bin", "prg.gfa", "update_DS"):
        assert open(f"{one}.{kind}.zip", "rb").read() == open(f"{four}.{kind}.zip", "rb").read(), kind
        with zipfile.ZipFile(f"{four}.{kind}.zip") as b:
            assert b.testzip() is None and len(b.namelist()) == len(files)


def test_more_ranks_than_alignments(tmp_path):
    """Three ranks, two alignments: a rank without files writes no segment and still takes part in the exchange."""
    d = tmp_path / "msas"
    d.mkdir()
    for s in range(2):
        (d / f"s{s}.fa").write_text(synth_config_fasta("B", s))
    env = dict(os.environ, MPRG_DIST_BACKEND="gloo", OMP_NUM_THREADS="1")
    one, three = tmp_path / "one" / "out", tmp_path / "three" / "out"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(d), str(one)], env=env)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                           "--master-addr", "127.0.0.1", "--master-port", "29539",
                           os.path.join(ROOT, "tests", "dist_worker.py"), str(d), str(three)], env=env)
    for name in ("out.prg.fa", "out.prg.bin.zip", "out.prg.gfa.zip", "out.update_DS.zip"):
        assert (tmp_path / "one" / name).read_bytes() == (tmp_path / "three" / name).read_bytes(), name
    assert len(list((tmp_path / "three").iterdir())) == 4


def test_segment_index_round_trip_and_merge(tmp_path):
    """The job's one collective carries segment INDEXES (utils/segments.py); the merge writes byte ranges of the segments in the
    run's order and one central directory per container — here on two hand-made segments, checked with zipfile."""
    import zlib
    from make_prg_amd.utils import segments
    from make_prg_amd.utils.zip_stream import StoredZipWriter
    idxs = []
    data = {0: {"b": b"BBBB", "d": b"D" * 70000}, 1: {"a": b"A", "c": b"CC"}}
    for r, members in data.items():
        prefix = str(tmp_path / f"o.rank{r}")
        fa = b"".join(b">" + k.encode() + b"\n" + v + b"\n" for k, v in members.items())
        open(prefix + ".prg.fa", "wb").write(fa)
        z = StoredZipWriter(prefix + ".prg.bin.zip")
        z.add_many([k + ".bin" for k in members], [[v] for v in members.values()], [zlib.crc32(v) for v in members.values()],
                   [len(v) for v in members.values()])
        z.close()
        idxs.append(dict(n=len(members), prefix=prefix, fa=[[k, len(k) + len(v) + 3] for k, v in members.items()],
                         zips={"bin": [[nb.decode(), crc, size, off] for nb, crc, size, off in z.entries]}))
        assert segments.unpack_index(segments.pack_index(idxs[-1])) == idxs[-1]
    n = segments.merge_segments(idxs, str(tmp_path / "o"), sort_key=lambda l: l + ".prg.fa")
    assert n == 4
    assert (tmp_path / "o.prg.fa").read_bytes() == b">a\nA\n>b\nBBBB\n>c\nCC\n>d\n" + b"D" * 70000 + b"\n"
    with zipfile.ZipFile(tmp_path / "o.prg.bin.zip") as z:
        assert z.testzip() is None and z.namelist() == ["a.bin", "b.bin", "c.bin", "d.bin"] and z.read("d.bin") == b"D" * 70000
    assert not (tmp_path / "o.rank0.prg.fa").exists() and not (tmp_path / "o.rank1.prg.bin.zip").exists()


def test_every_rank_places_its_own_bytes(tmp_path):
    """The multi-rank form of the merge: every rank calls it with all indexes; rank 0 creates the files and writes the directories,
    each rank copies only ITS segment's ranges (here the two ranks one after the other, the barrier a no-op: the ranges are
    disjoint, so the order does not matter) — same files as the one-caller form."""
    import shutil
    import zlib
    from make_prg_amd.utils import segments
    from make_prg_amd.utils.zip_stream import StoredZipWriter
    data = {0: {"a": b"A" * 5, "b": b"BBBB"}, 1: {"c": b"CC", "d": b"D" * 70000}}

    def make(root):
        idxs = []
        for r, members in data.items():
            prefix = str(root / f"seg.rank{r}")
            open(prefix + ".prg.fa", "wb").write(b"".join(b">" + k.encode() + b"\n" + v + b"\n" for k, v in members.items()))
            z = StoredZipWriter(prefix + ".prg.gfa.zip")
            z.add_many([k + ".gfa" for k in members], [[v] for v in members.values()], [zlib.crc32(v) for v in members.values()],
                       [len(v) for v in members.values()])
            z.close()
            idxs.append(dict(n=len(members), prefix=prefix, fa=[[k, len(k) + len(v) + 3] for k, v in members.items()],
                             zips={"gfa": [[nb.decode(), crc, size, off] for nb, crc, size, off in z.entries]}))
        return idxs

    (tmp_path / "one").mkdir(); (tmp_path / "two").mkdir()
    segments.merge_segments(make(tmp_path / "one"), str(tmp_path / "one" / "o"), sort_key=lambda l: l + ".prg.fa")
    idxs = make(tmp_path / "two")
    calls = []
    for r in (0, 1):
        n = segments.merge_segments(idxs, str(tmp_path / "two" / "o"), sort_key=lambda l: l + ".prg.fa", rank=r, world=2,
                                    barrier=lambda: calls.append(r))
        assert n == 4
        assert not os.path.exists(idxs[r]["prefix"] + ".prg.fa")            # a rank removes its own segments ...
        assert r == 1 or os.path.exists(idxs[1]["prefix"] + ".prg.fa")      # ... and leaves the others'
    assert calls == [0, 0, 1, 1]
    for name in ("o.prg.fa", "o.prg.gfa.zip"):
        assert (tmp_path / "one" / name).read_bytes() == (tmp_path / "two" / name).read_bytes(), name
    with zipfile.ZipFile(tmp_path / "two" / "o.prg.gfa.zip") as z:
        assert z.testzip() is None and z.namelist() == ["a.gfa", "b.gfa", "c.gfa", "d.gfa"]


def test_shards_are_contiguous_in_the_run_order(tmp_path):
    """A rank's shard is a contiguous stretch of the run's sorted loci (so that its part of every output file is one byte range),
    balanced by file size to within one file."""
    from make_prg_amd.pipeline import sort_key
    from make_prg_amd.subcommands.from_msa import shard_files
    import random
    rnd = random.Random(5)
    files = []
    for i in range(200):
        p = tmp_path / f"g{rnd.randrange(10 ** 6):06d}_{i}.fa"
        p.write_text("x" * rnd.randrange(100, 3000))
        files.append(p)
    order = sorted(files, key=sort_key)
    for world in (2, 3, 8):
        parts = [shard_files(files, r, world) for r in range(world)]
        assert [f for part in parts for f in part] == order          # contiguous, in order, complete
        loads = [sum(f.stat().st_size for f in part) for part in parts]
        assert max(loads) - min(loads) <= 2 * 3000, loads


def test_shards_are_disjoint_and_complete(tmp_path):
    from make_prg_amd.subcommands.from_msa import shard_files
    files = []
    for i in range(11):
        p = tmp_path / f"f{i}.fa"
        p.write_text("x" * (100 * (i + 1)))
        files.append(p)
    parts = [shard_files(files, r, 3) for r in range(3)]
    assert sorted(p for part in parts for p in part) == sorted(files)
    sizes = [sum(p.stat().st_size for p in part) for part in parts]
    assert max(sizes) - min(sizes) <= 1100

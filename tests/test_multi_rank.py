"""N > 1 path: two ranks (gloo, CPU, emulation backend) shard a directory of alignments, rank 0 gathers and writes;
the result must equal the single-process run byte for byte (.prg.fa) and member for member (zips)."""
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
    for kind in ("bin", "gfa"):
        with zipfile.ZipFile(f"{one}.prg.{kind}.zip") as a, zipfile.ZipFile(f"{two}.prg.{kind}.zip") as b:
            assert sorted(a.namelist()) == sorted(b.namelist()) and len(a.namelist()) == 9
            for n in a.namelist():
                assert a.read(n) == b.read(n)


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
        with zipfile.ZipFile(f"{one}.{kind}.zip") as a, zipfile.ZipFile(f"{four}.{kind}.zip") as b:
            assert sorted(a.namelist()) == sorted(b.namelist()) and len(a.namelist()) == len(files)
            if kind != "update_DS":
                for n in a.namelist():
                    assert a.read(n) == b.read(n)


def test_record_packing_round_trip():
    from make_prg_amd.subcommands.from_msa import pack_records, unpack_records
    local = {"b": dict(prg="AC 5 G 6 T 5 ", pickle=b"\x80\x04", bin=b"\x01\x00\x00\x00", gfa=b"H\tVN\n"),
             "a": dict(prg="", bin=b""), "c": dict(prg="ACGT")}
    assert unpack_records(pack_records(local)) == local
    assert unpack_records(pack_records({})) == {}


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

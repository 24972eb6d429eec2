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

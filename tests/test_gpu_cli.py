"""The from_msa command line on the GPU box, in a child process, with `-t 3` host worker processes sharing the device:
the .prg.fa it writes must hold the oracle's PRG of every locus.  (Runs the CLI as a subprocess so that its forked
workers never descend from a process that has initialised the GPU.)"""
import os
import subprocess
import sys

import pytest

import oracle.from_msa_oracle as orc
from make_prg_amd.utils.synthetic import synth_config_fasta

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("backend", ["runtime", "torch"])
def test_command_line_with_host_workers(tmp_path, backend):
    d = tmp_path / "msas"
    d.mkdir()
    want = {}
    for seed in range(500, 512):
        text = synth_config_fasta("B", seed)
        (d / f"gene{seed}.fa").write_text(text)
        want[f"gene{seed}"] = orc.build_locus_from_text(text, 5, 7)[0]
    prefix = tmp_path / "out" / "pan"
    env = dict(os.environ, PYTHONPATH=ROOT, MPRG_BACKEND=backend)
    res = subprocess.run([sys.executable, "-m", "make_prg_amd", "from_msa", "-i", str(d), "-o", str(prefix), "-t", "3", "-O", "p"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    got = {}
    lines = (tmp_path / "out" / "pan.prg.fa").read_text().splitlines()
    for name, prg in zip(lines[0::2], lines[1::2]):
        got[name[1:]] = prg
    assert got == want


def test_streamed_pipeline_equals_the_object_path_on_the_gpu(tmp_path):
    """`-O a` through the streamed pipeline (several chunks, runtime backend, native threads) and through the per-locus object
    path on the same directory — a gzipped file and a file with duplicate row ids among them: same .prg.fa bytes, same zip
    members (CRCs checked), equal builders in update_DS.zip."""
    import gzip
    import zipfile
    from make_prg_amd.prg_builder import PrgBuilderZipDatabase
    d = tmp_path / "msas"
    d.mkdir()
    for seed in range(700, 716):
        text = synth_config_fasta("B", seed)
        if seed % 5 == 1:
            with gzip.open(d / f"gene{seed}.fa.gz", "wt") as fh:
                fh.write(text)
        else:
            (d / f"gene{seed}.fa").write_text(text)
    (d / "dup_ids.fa").write_text(">a\nACGTACGTACGTTTTT\n>a\nACGTACGAACGTTTTT\n>b\nACGTACGTACGTATTT\n")
    outs = {}
    for mode in ("1", "0"):
        prefix = tmp_path / f"out{mode}" / "pan"
        env = dict(os.environ, PYTHONPATH=ROOT, MPRG_PIPELINE=mode, MPRG_CHUNK="5")
        res = subprocess.run([sys.executable, "-m", "make_prg_amd", "from_msa", "-i", str(d), "-o", str(prefix), "-t", "4", "-O", "a"],
                             cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        files = {p.name: p for p in (tmp_path / f"out{mode}").iterdir()}
        assert sorted(files) == ["pan.prg.bin.zip", "pan.prg.fa", "pan.prg.gfa.zip", "pan.update_DS.zip"]
        got = {"fa": files["pan.prg.fa"].read_bytes()}
        for kind in ("bin", "gfa"):
            with zipfile.ZipFile(files[f"pan.prg.{kind}.zip"]) as z:
                assert z.testzip() is None
                got[kind] = {m: z.read(m) for m in sorted(z.namelist())}
        db = PrgBuilderZipDatabase(files["pan.update_DS.zip"])
        db.load()
        got["builders"] = {l: db.get_PrgBuilder(l) for l in db.get_loci_names()}
        db.close()
        outs[mode] = got
    a, b = outs["1"], outs["0"]
    assert a["fa"] == b["fa"] and a["bin"] == b["bin"] and a["gfa"] == b["gfa"]
    assert sorted(a["builders"]) == sorted(b["builders"]) and len(a["builders"]) == 17
    for l in a["builders"]:
        assert a["builders"][l].build_prg() == b["builders"][l].build_prg(), l


def test_two_ranks_on_the_gpu_write_the_single_rank_files(tmp_path):
    """`torchrun` with two ranks (gloo for the index exchange; both ranks on the one GPU of the box — RCCL refuses two ranks on one
    device) against the plain command line: every rank streams its shard into segment files on the device, rank 0 merges them;
    all four output files byte-identical, a gzipped file and a locus the curation policy skips among the inputs."""
    import gzip
    d = tmp_path / "msas"
    d.mkdir()
    for seed in range(800, 840):
        text = synth_config_fasta("B" if seed % 3 else "C", seed)
        if seed % 7 == 1:
            with gzip.open(d / f"gene{seed}.fa.gz", "wt") as fh:
                fh.write(text)
        else:
            (d / f"gene{seed}.fa").write_text(text)
    (d / "bad.fa").write_text(">a\nACGTNNNNNNNNACGT\n>b\nACGANNNNNNNNACGT\n>c\nACGANNNNNNNNACGA\n")
    env = dict(os.environ, PYTHONPATH=ROOT, MPRG_CHUNK="8", MPRG_DIST_BACKEND="gloo")
    one, two = tmp_path / "one" / "pan", tmp_path / "two" / "pan"
    args = ["from_msa", "-i", str(d), "-t", "4", "-O", "a"]
    res = subprocess.run([sys.executable, "-m", "make_prg_amd"] + args + ["-o", str(one)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", "-m", "make_prg_amd"] + args + ["-o", str(two)], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    names = sorted(p.name for p in (tmp_path / "one").iterdir())
    assert names == ["pan.prg.bin.zip", "pan.prg.fa", "pan.prg.gfa.zip", "pan.update_DS.zip"] == sorted(p.name for p in (tmp_path / "two").iterdir())
    for n in names:
        assert (tmp_path / "one" / n).read_bytes() == (tmp_path / "two" / n).read_bytes(), n
    assert (tmp_path / "one" / "pan.prg.fa").read_text().count(">") >= 39


def test_one_rank_under_rccl_writes_the_single_process_files(tmp_path):
    """The multi-rank path of the command line with RCCL under it, as far as ONE GPU allows: `torchrun --nproc-per-node 1` with
    MPRG_DIST_FORCE=1 makes the `nccl` process group for the one rank — segments, the index all-gather on DEVICE tensors
    (allgather_bytes), the one-HIP-runtime check with a live communicator, placement — and the four output files equal the plain run's."""
    d = tmp_path / "msas"
    d.mkdir()
    for seed in range(860, 890):
        (d / f"gene{seed}.fa").write_text(synth_config_fasta("B" if seed % 3 else "C", seed))
    env = dict(os.environ, PYTHONPATH=ROOT, MPRG_CHUNK="8")
    env.pop("MPRG_DIST_BACKEND", None)
    one, two = tmp_path / "one" / "pan", tmp_path / "two" / "pan"
    args = ["from_msa", "-i", str(d), "-t", "4", "-O", "a"]
    res = subprocess.run([sys.executable, "-m", "make_prg_amd"] + args + ["-o", str(one)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29549", "-m", "make_prg_amd"] + args + ["-o", str(two)], cwd=ROOT,
                         env=dict(env, MPRG_DIST_FORCE="1"), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "index exchange" in res.stderr or "index exchange" in res.stdout, (res.stderr[-1500:], res.stdout[-500:])   # (run_ranks ran)
    names = sorted(p.name for p in (tmp_path / "one").iterdir())
    assert names == ["pan.prg.bin.zip", "pan.prg.fa", "pan.prg.gfa.zip", "pan.update_DS.zip"] == sorted(p.name for p in (tmp_path / "two").iterdir())
    for n in names:
        assert (tmp_path / "one" / n).read_bytes() == (tmp_path / "two" / n).read_bytes(), n


def test_command_line_on_one_deep_alignment(tmp_path):
    """The command line on the FASTA file of the parity-checked deep alignment (tests/golden/ddeep.json: 2 000 x 4 000, -N 7; the real
    reference's PRG): levels with big clustering problems take the multi-workgroup forms; every output type is written."""
    import hashlib
    import json
    from make_prg_amd.utils.synthetic import synth_deep_fasta
    with open(os.path.join(ROOT, "tests", "golden", "ddeep.json")) as fh:
        g = json.load(fh)
    d = tmp_path / "msas"
    d.mkdir()
    (d / "ddeep.fa").write_text(synth_deep_fasta(g["seed"], g["S"], g["C"]))
    prefix = tmp_path / "out" / "deep"
    res = subprocess.run([sys.executable, "-m", "make_prg_amd", "from_msa", "-i", str(d), "-o", str(prefix), "-N", str(g["N"]), "-L", str(g["L"]),
                          "-t", "2", "-O", "a"], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = (tmp_path / "out" / "deep.prg.fa").read_text().splitlines()
    assert len(lines) == 2 and len(lines[1]) == g["expect"]["prg_len"]
    assert hashlib.sha256(lines[1].encode()).hexdigest() == g["expect"]["prg_sha256"]
    for suffix in (".prg.bin", ".prg.gfa", ".update_DS.zip"):
        assert (tmp_path / "out" / ("deep" + suffix)).stat().st_size > 0

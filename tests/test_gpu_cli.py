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

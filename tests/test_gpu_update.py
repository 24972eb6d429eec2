"""BASELINE.json config 5 on the MI355X: `update` (our from_msa -> update_DS -> our update with the recorded MAFFT
answers; the re-entries of all touched leaves in one resident batch through the HIP kernels) against what the real
reference produced for its own ten update cases incl. sample_example (tests/golden/update.json.gz): byte-equal
.prg.fa / .bin / .gfa, tree-equal update_DS (node ids, kinds, levels, per-node alignments, prg_index, counters)."""
import json
import os
import subprocess
import sys

import pytest

from tests import update_common as uc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_update_cases_on_gpu(tmp_path):
    from make_prg_amd import device
    from make_prg_amd.backend import HipBackend
    hip = HipBackend(0)
    device.set_backend(hip)
    try:
        golden = uc.load_cases()
        n = 0
        for case in golden["cases"]:
            n += uc.check_outputs(case, uc.run_case(case, tmp_path, backend=hip))
        assert len(golden["cases"]) == 10 and n >= 20
    finally:
        device.set_backend(None)


def test_update_command_line_sample_example(tmp_path):
    """The two sub-commands as a user runs them: from_msa on sample_example's alignments, then update with
    --aligner-replay (child processes; `-t 2` host workers / aligner threads)."""
    case = next(c for c in uc.load_cases()["cases"] if c["case"] == "sample_example_update")
    src = tmp_path / "msas"
    src.mkdir()
    for f in case["inputs"]:
        (src / f["name"]).write_text(f["fasta"])
    (tmp_path / "denovo_paths.txt").write_text(case["denovo_paths"])
    (tmp_path / "replay.json").write_text(json.dumps(case["aligner_replay"]))
    env = dict(os.environ, PYTHONPATH=ROOT)
    base, out = str(tmp_path / "base" / "sample"), str(tmp_path / "out" / "sample_example_update")
    for args in (["from_msa", "-i", str(src), "-o", base, "-t", "2"],
                 ["update", "-u", base + ".update_DS.zip", "-d", str(tmp_path / "denovo_paths.txt"), "-o", out, "-t", "2",
                  "-D", str(case["long_deletion_threshold"]), "--aligner-replay", str(tmp_path / "replay.json")]):
        res = subprocess.run([sys.executable, "-m", "make_prg_amd"] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                             timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
    assert uc.check_outputs(case, out) == 3

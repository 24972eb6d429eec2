"""BASELINE.json config C at its full size on the MI355X: all 30 000 alignments of the synthetic pan-genome
(~100 x 1-3 kb each, seeds 0..29999, -N 5 -L 7) through the HIP path, every PRG and node count against the digest
fixture the oracle produced (tests/golden/config_c_digests.bin).  Runs in a child process (tests/config_c_full.py) so
that its generator workers are forked from a process that has not touched the GPU, whatever ran before in this one."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_all_30000_config_c_loci_match_the_oracle_digests():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "config_c_full.py"), "0", "30000", "7500"],
                         cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=1500)
    assert res.stdout.strip(), res.stderr[-3000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert res.returncode == 0 and out["mismatches"] == 0, (out, res.stderr[-2000:])
    assert out["loci"] == 30000 and out["kmeans_fits"] > 1_000_000

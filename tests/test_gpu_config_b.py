"""BASELINE.json config B at full size on the GPU: 1 000 synthetic alignments (50 x 500, seeds 0..999, -N 5 -L 7) in
ONE resident batch, every PRG compared with the oracle (run in worker processes before the GPU is touched), plus a
config-C sample.  Size-independent checks on top: every PRG round-trips through the binary encoder and its site
markers pair up."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(args):
    cfg, seed = args
    import oracle.from_msa_oracle as orc
    from make_prg_amd.utils.synthetic import synth_config_fasta
    prg, b, root = orc.build_locus_from_text(synth_config_fasta(cfg, seed), 5, 7)
    return prg, b.next_node_id


def _expected(cfg, seeds):
    import oracle.from_msa_oracle as orc
    orc.build_kmeans_lib()
    with mp.get_context("fork").Pool(min(os.cpu_count() or 1, 64)) as pool:
        return pool.map(_oracle, [(cfg, s) for s in seeds], chunksize=4)


CASES = [("B", 0, 1000), ("C", 5000, 5128)]


@pytest.fixture(scope="module")
def oracle_results():
    """All expectations first: the worker processes are forked before this process makes its first HIP call
    (this file sorts before the other GPU test files)."""
    return {(cfg, lo, hi): _expected(cfg, list(range(lo, hi))) for cfg, lo, hi in CASES}


def run_gpu(cfg, seeds):
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import ForestEngine
    from make_prg_amd.msa import load_alignment_text
    from make_prg_amd.utils.synthetic import synth_config_fasta
    eng = ForestEngine(HipBackend(0), 5, 7)
    eng.load([load_alignment_text(synth_config_fasta(cfg, s)) for s in seeds])
    eng.run_forest()
    prgs = eng.assemble_prgs()
    n_nodes = np.bincount(eng.tab["msa"], minlength=len(seeds))
    return prgs, n_nodes


def check_markers(prg):
    units = prg.split()
    markers = [int(u) for u in units if u.isdigit()]
    odd = [m for m in markers if m % 2 == 1]
    assert all(odd.count(m) == 2 for m in set(odd)), "every site opens and closes once"
    assert all(u for u in units)


@pytest.mark.parametrize("cfg,lo,hi", CASES)
def test_full_config_against_oracle(oracle_results, cfg, lo, hi):
    seeds = list(range(lo, hi))
    want = oracle_results[(cfg, lo, hi)]
    prgs, n_nodes = run_gpu(cfg, seeds)
    bad = [s for s, p, (w, _) in zip(seeds, prgs, want) if p != w]
    assert not bad, f"{len(bad)} of {len(seeds)} loci differ from the oracle, first seeds {bad[:5]}"
    assert [int(x) for x in n_nodes] == [n for _, n in want]
    for p in prgs[::25]:
        check_markers(p)

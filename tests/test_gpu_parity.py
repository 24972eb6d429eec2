"""Parity tests proper: the HIP path (libmprg_hip.so through the C ABI, device buffers in HBM) against the golden
vectors of the real reference and against the oracle.  Run on the MI355X box with `-m gpu`."""
import numpy as np
import pytest

from tests import parity_common as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["torch", "runtime"])
def hip(request):
    """Both product backends over the same library: device buffers and streams from torch, or from the library's own
    mprg_rt_* plumbing (no torch; what the command line uses).  Either raises if the library or the GPU is missing."""
    from make_prg_amd.backend import HipBackend, HipRuntimeBackend
    return HipBackend(0) if request.param == "torch" else HipRuntimeBackend(0)


def test_library_is_the_hip_build(hip):
    assert b"hip gfx950" in hip.lib.mprg_version()
    assert hip.n_cus >= 200


def test_integration_cases(hip, golden_integration):
    assert pc.check_integration(hip, golden_integration) >= 30


def test_synthetic_config_b(hip, golden_synthetic):
    assert pc.check_synthetic(hip, golden_synthetic, configs=("B",)) == 40


def test_synthetic_config_c_and_deep(hip, golden_synthetic):
    assert pc.check_synthetic(hip, golden_synthetic, configs=("C", "Dsmall")) == 7


def test_round_lists_on_side_streams(hip, golden_synthetic, monkeypatch):
    """forest.KM_SIDE_STREAMS (what bench.py switches on for a rank with one host worker): a round's launch lists forked to side
    streams and joined; same trees, same PRGs."""
    import make_prg_amd.forest as forest
    monkeypatch.setattr(forest, "KM_SIDE_STREAMS", True)
    assert pc.check_synthetic(hip, golden_synthetic, configs=("B", "C")) >= 40


def test_kmeans_forms_of_earlier_rounds(hip, golden_synthetic, monkeypatch):
    """The default since round 6 is the LDS form of the fits (forest.KM_MODE bit 2).  The small / general workgroup forms it replaced
    (KM_MODE = 2) stay entry points of the ABI: fused and per round, same trees and PRGs."""
    import make_prg_amd.forest as F
    monkeypatch.setattr(F, "KM_MODE", 2)
    monkeypatch.setattr(F, "KM_LDS_ENTRY", "mprg_kmeans_fit_wave")
    monkeypatch.setattr(F, "KM_LISTS", ((F.KM_LDS_ENTRY, 0), (F.KM_LDS_ENTRY, 1), (F.KM_LDS_ENTRY, 2), (F.KM_LDS_ENTRY, 3), ("mprg_kmeans_fit", None),
                                        ("mprg_kmeans_fit_small", 0), ("mprg_kmeans_fit_small", 1)))
    for loop in ("fused", "rounds"):
        monkeypatch.setattr(F, "KLOOP", loop)
        assert pc.check_synthetic(hip, golden_synthetic, configs=("B", "C")) >= 40


def test_diagnostic_build_runs_the_fused_loops():
    """The diagnostic build of the kernels (-DKM_PHASE_TIMING, make_prg_amd/_lib/libmprg_hip_timing.so: clock marks between the phases of
    a fit) with the FUSED clustering loop on the LDS form, in a process of its own, against the oracle.  (Round 5's diagnostic build
    faulted in the fused small form, k_cluster_loop_small — profiles/r06/NOTES.md; the LDS loop that replaced it must not.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "make_prg_amd", "_lib", "libmprg_hip_timing.so")
    if not os.path.exists(lib):
        pytest.skip("the diagnostic library is not built (tools/phase_timing.py says how)")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests import parity_common as pc\n"
            "from tests.random_msas import random_cases\n"
            "from make_prg_amd.backend import HipRuntimeBackend\n"
            "from make_prg_amd.utils.synthetic import synth_config_fasta\n"
            "pc.ENGINE = 'forest'\n"
            "be = HipRuntimeBackend(0)\n"
            "assert b'hip gfx950' in be.lib.mprg_version()\n"
            "eng = pc.check_vs_oracle(be, random_cases(61, 60) + [synth_config_fasta('C', s) for s in range(900, 912)], 5, 7)\n"
            "assert eng.kloop_fused and eng.counters['fits'] > 300\n" % root)
    res = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, MPRG_HIP_LIB=lib, MPRG_KLOOP="fused"), capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-1500:])


def test_compact_columns(hip):
    """A8 on the device (mprg_compact_columns); the tree dumps of the tests above go through it too (node.alignment)."""
    assert pc.check_compact_columns(hip) == 6


def test_kmeans_known_answers(hip, golden_kmeans):
    """The device KMeans against the scikit-learn answers captured from the reference run (labels, inertia bits)."""
    from tests.kmeans_direct import run_kmeans_fits
    fits = golden_kmeans["fits"]
    got = run_kmeans_fits(hip, fits)
    for g, f in zip(got, fits):
        assert g["labels"] == f["labels"]
        assert g["inertia_hex"] == f["inertia"]
        assert g["n_iter"] == f["n_iter"]


def test_kmeans_known_answers_persistent_workgroups(hip, golden_kmeans):
    """The same scikit-learn answers through mprg_kmeans_fit (persistent workgroups, per-restart arrays in scratch slots,
    selection fused), with a few slots (many fits per workgroup) and with as many slots as fits."""
    from tests.kmeans_direct import run_kmeans_fits
    fits = golden_kmeans["fits"]
    # one-launch: no scratch slots, a workgroup per fit; wave: a wavefront per fit; split / wide: a workgroup (64 / 1 024 threads) per
    # restart + a selection launch
    for path, slots in (("one-launch", 0), ("wave", 0), ("small", 0), ("lds", 0), ("split", 0), ("wide", 0), ("wide-bytes", 0), ("wide-stats", 0), ("fit", 5), ("fit", 4096)):
        got = run_kmeans_fits(hip, fits, path=path, n_slots=slots)
        for g, f in zip(got, fits):
            assert g["labels"] == f["labels"]
            assert g["inertia_hex"] == f["inertia"]
            assert g["n_iter"] == f["n_iter"]


def test_argpartition_with_median_of_medians_fallback(hip):
    """np.argpartition of the relocation on adversarial inputs (answers of the real NumPy, generic introselect)."""
    from tests.test_argpartition import check_device
    assert check_device(hip) == 9


def test_kmer_sizes_above_16(hip):
    """-L 17 / 20 / 24 / 33: the real reference's answers (verified hash keys in the k-mer dictionary)."""
    from tests.long_kmer_common import check_long_kmers
    assert check_long_kmers(hip) == 4


def test_batch_of_fresh_seeds_against_oracle(hip):
    from make_prg_amd.utils.synthetic import synth_config_fasta
    texts = [synth_config_fasta("B", s) for s in range(100, 148)]
    pc.check_vs_oracle(hip, texts)


def test_edge_cases_against_oracle(hip):
    from tests.edge_cases import EDGE_FASTAS
    for N, L, texts in EDGE_FASTAS:
        pc.check_vs_oracle(hip, texts, N, L)


def test_random_small_alignments_against_oracle(hip):
    from tests.random_msas import random_cases
    for N, L, seed in [(5, 7, 31), (5, 3, 32), (2, 1, 33), (5, 2, 34)]:
        pc.check_vs_oracle(hip, random_cases(seed, 200), N, L)


def test_node_object_host_on_gpu(hip, golden_integration, monkeypatch):
    monkeypatch.setattr(pc, "ENGINE", "nodes")
    assert pc.check_integration(hip, golden_integration) >= 30


def test_empty_cluster_relocation_on_gpu(hip):
    from tests.test_kmeans_relocation import check, degenerate_fits
    fits = degenerate_fits(8, 80)
    check(hip, fits)
    check(hip, fits, path="one-launch")
    check(hip, fits, path="fit", n_slots=3)
    check(hip, fits, path="fit", n_slots=1024)


def test_batched_reentry_below_existing_nodes_on_gpu(hip):
    """NodeFactory.build_many (the GPU part of the reference's `update`): every touched leaf of every locus in one
    resident batch, sub-trees and node ids against vectors of the real reference."""
    import gzip
    import json
    import os
    from make_prg_amd import device
    from make_prg_amd.recursion_tree import NodeFactory
    from tests import test_reentry_emulated as tr
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reentry.json.gz"), "rt") as fh:
        cases = json.load(fh)["cases"]
    device.set_backend(hip)
    try:
        builders = [tr._builder(c, c["first_node_id"]) for c in cases]
        jobs = [job for c, b in zip(cases, builders) for job in tr._jobs(c, b)]
        subs = NodeFactory.build_many(jobs)
        k = 0
        for c, b in zip(cases, builders):
            for j in c["jobs"]:
                assert tr._dump(subs[k]) == j["subtree"], c["file"]
                k += 1
            assert b.next_node_id == c["jobs"][-1]["next_node_id"]
    finally:
        device.set_backend(None)


def test_exported_helper_functions_on_gpu(hip):
    """cluster_sequences' exported helpers and NodeFactory's private helpers against the real reference's answers, the
    one-reference-like test on the HIP kernel of the recursion (tests/golden/helpers.json.gz)."""
    from make_prg_amd import device
    from tests.test_helpers_api import check_all, load
    device.set_backend(hip)
    try:
        assert check_all(load()) > 300
    finally:
        device.set_backend(None)


def test_kmer_dictionary_and_counts_by_many_workgroups(hip, golden_integration, monkeypatch):
    """mprg_kmer_dictionary_parts / mprg_kmer_counts_parts — a problem's k-mer occurrences shared by many workgroups through global
    atomics (what the top levels of one deep alignment take) — forced for EVERY level here (KD_PARTS_FROM = 1; every level counted
    as big, so the counts go through the parts form too), on the device's real atomics: same ids, hence the same column order of
    every count matrix and the same KMeans sums as the one-workgroup kernels; answers of the oracle and the real reference's goldens."""
    import make_prg_amd.forest as forest
    from tests.random_msas import random_cases
    monkeypatch.setattr(forest, "KD_PARTS_FROM", 1)
    monkeypatch.setattr(forest, "KM_BIG_BYTES", 1)
    monkeypatch.setattr(pc, "ENGINE", "forest")
    calls = []
    orig = hip.call
    monkeypatch.setattr(hip, "call", lambda name, *a, **k: (calls.append(name), orig(name, *a, **k))[1])
    pc.check_vs_oracle(hip, random_cases(61, 60), 5, 7)
    pc.check_vs_oracle(hip, random_cases(62, 30), 3, 3)
    assert pc.check_integration(hip, golden_integration) >= 30
    assert calls.count("mprg_kmer_dictionary_parts") >= 3 and calls.count("mprg_kmer_counts_parts") >= 3 and "mprg_kmer_dictionary" not in calls


def test_sample_tables_by_tiles_equal_the_chains_by_threads(hip):
    """K6's sample-sample tables of the global form (k_kmeans_prepare_tables_tiled: all chains of a sample pair advanced together over
    LDS-staged features) are the doubles of the thread-per-element kernel: every block boundary of the CPU kernels' K loops, and problems
    of the size a deep alignment's levels hold."""
    from tests.kmeans_tables import SHAPES, check_lds_tables, check_tiled_tables
    assert check_tiled_tables(hip, SHAPES + [(300, 3000), (700, 4100), (130, 6200)]) == []
    assert check_lds_tables(hip) == []          # K6's LDS form: a thread per pair over the matrix in LDS

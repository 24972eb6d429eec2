"""Host engine + kernel LOGIC on the CPU emulation build of the kernel source (tests/emu) against the golden
vectors.  This is not the product path (the product path is HIP only, tests/test_gpu_parity.py); it is here so the
host recursion driver and every kernel body are exercised in the GPU-less container."""
import numpy as np
import pytest

from tests.emu.backend import EmuBackend
from tests import parity_common as pc


@pytest.fixture(scope="module")
def emu():
    return EmuBackend()


def test_integration_cases(emu, golden_integration):
    assert pc.check_integration(emu, golden_integration) >= 30


def test_synthetic_b(emu, golden_synthetic):
    assert pc.check_synthetic(emu, golden_synthetic, configs=("B",)) == 40


def test_synthetic_c_and_deep(emu, golden_synthetic):
    assert pc.check_synthetic(emu, golden_synthetic, configs=("C", "Dsmall")) == 7


def test_node_object_host_matches_too(emu, golden_integration, golden_synthetic, monkeypatch):
    """engine.BatchEngine (per-node bookkeeping, used by the PrgBuilder / NodeFactory API) on the same goldens."""
    monkeypatch.setattr(pc, "ENGINE", "nodes")
    assert pc.check_integration(emu, golden_integration) >= 30
    assert pc.check_synthetic(emu, golden_synthetic, configs=("B",), limit=12) == 12


def test_load_time_consensus_counts_from_the_device(emu):
    """mprg_column_residue_counts + the host's seeded choice == the host-only majority consensus (reference
    utils/seq_utils.py:246-290), on alignments with many N, ties, lower case, ambiguity codes and all-N columns."""
    import numpy as np
    from make_prg_amd.engine import BatchEngine
    from make_prg_amd.msa import load_alignment_text
    rng = np.random.default_rng(12)
    texts = []
    for t in range(60):
        S, C = int(rng.integers(1, 40)), int(rng.integers(1, 300))
        alphabet = np.frombuffer(b"ACGTacgtNNNn--RYKMSW" if t % 3 else b"ACGTNN-", np.uint8)
        rows = alphabet[rng.integers(0, len(alphabet), (S, C))]
        if t % 5 == 0:
            rows[:, int(rng.integers(0, C))] = ord("N")                 # a column of nothing but N
        texts.append("".join(f">r{i} d\n{bytes(r).decode()}\n" for i, r in enumerate(rows)))
    want = [load_alignment_text(t) for t in texts]
    got = [load_alignment_text(t, defer_n=True) for t in texts]
    assert any(m.pending_n for m in got)
    BatchEngine(emu, 5, 7).load(got)
    for w, g in zip(want, got):
        assert not g.pending_n and np.array_equal(w.data, g.data)


def test_kmer_sizes_above_16(emu):
    from tests.long_kmer_common import check_long_kmers
    assert check_long_kmers(emu) == 4


def test_compact_columns_matches_numpy(emu):
    """mprg_compact_columns (A8) behind remove_columns_full_of_gaps_from_MSA (tests/parity_common.py)."""
    assert pc.check_compact_columns(emu) == 6

"""`make_prg_amd update` (BASELINE.json config 5) on the CPU emulation build against the real reference's results for
its own ten update test cases (tests/golden/update.json.gz), plus the denovo_paths parser and align() known answers."""
import pytest

from make_prg_amd import device
from tests import update_common as uc
from tests.emu.backend import EmuBackend


@pytest.fixture(scope="module")
def golden():
    return uc.load_cases()


@pytest.fixture(autouse=True, scope="module")
def _emu():
    device.set_backend(EmuBackend())
    yield
    device.set_backend(None)


def test_reference_update_cases(golden, tmp_path):
    n = 0
    for case in golden["cases"]:
        prefix = uc.run_case(case, tmp_path, backend=device.get_backend())
        n += uc.check_outputs(case, prefix)
    assert len(golden["cases"]) == 10 and n >= 20


def test_align_known_answers(golden):
    from make_prg_amd.utils.seq_utils import align
    for a, b, want in golden["meta"]["align_known_answers"]:
        assert list(align(a, b)) == want
    assert align("A", "T", 0, 0, 0, 0) == ("A-", "-T")          # several equally good alignments: the first one


def test_variant_spanning_two_leaves_is_split():
    """A variant whose ref crosses a leaf boundary becomes one sub-variant per ML path node (reference
    update/denovo_variants.py:183-286)."""
    from make_prg_amd.update.denovo_variants import DenovoLocusInfo, DenovoVariant
    from make_prg_amd.update.ml_path import MLPath, MLPathNode
    path = MLPath([MLPathNode((0, 4), "ACGT"), MLPathNode((10, 14), "TTGA")])
    info = DenovoLocusInfo("s", "l", path, [DenovoVariant(2, "GTTT", "GCTA")])
    got = [(u.ml_path_node_key, u.new_node_sequence) for u in info.get_update_data()]
    assert got == [((0, 4), "ACGC"), ((10, 14), "TAGA")]
    ins = DenovoLocusInfo("s", "l", path, [DenovoVariant(8, "", "CC")])        # after the last base of the last node
    assert [(u.ml_path_node_key, u.new_node_sequence) for u in ins.get_update_data()] == [((10, 14), "TTGACC")]


def test_replay_aligner_never_guesses():
    from make_prg_amd.msa import MSA
    from make_prg_amd.utils.msa_aligner import ReplayAligner
    with pytest.raises(KeyError):
        ReplayAligner([]).get_updated_alignment(MSA.from_strings(["ACGT"]), {"ACGA"})

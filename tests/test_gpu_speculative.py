"""Levels without host waits on the MI355X (mprg_forest_level): the second and third forest of a resident batch against the first
(per-step host) and against the oracle; capacities that are too small fall back.  Same checks as tests/test_speculative_emulated.py,
through the HIP library."""
import pytest

from make_prg_amd.forest import ForestEngine
from make_prg_amd.msa import load_alignment_text
from make_prg_amd.utils.synthetic import synth_config_fasta
from tests.random_msas import random_cases
from tests.test_speculative_emulated import dump, reset

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["torch", "runtime"])
def hip(request):
    from make_prg_amd.backend import HipBackend, HipRuntimeBackend
    return HipBackend(0) if request.param == "torch" else HipRuntimeBackend(0)


@pytest.fixture(scope="module")
def texts(golden_integration):
    t = [synth_config_fasta("B", s) for s in range(40)] + [synth_config_fasta("C", s) for s in range(60)] + random_cases(5, 100)
    t += [l["fasta"] for c in golden_integration["cases"] if c["case"] in ("fails_2", "contains_n") and (c["N"], c["L"]) == (5, 7) for l in c["loci"]]
    return t


def test_forests_from_a_plan_equal_the_first_and_the_oracle(hip, texts):
    import oracle.from_msa_oracle as orc
    eng = ForestEngine(hip, 5, 7)
    eng.load([load_alignment_text(t) for t in texts])
    eng.run_forest()
    first = dump(eng, len(texts))
    for i in (0, 39, 40, 99, 150):
        assert first[0][i] == orc.build_locus_from_text(texts[i], 5, 7)[0]
    for _ in range(3):
        reset(eng)
        eng.run_forest()
        assert eng.counters["syncs"] == 1 and eng.counters.get("plan_misses", 0) == 0
        assert dump(eng, len(texts)) == first


def test_planned_forests_with_the_general_form_beside_the_lds_classes(hip, texts, monkeypatch):
    """forest.KM_SIDE_STREAMS in a forest enqueued without waits (what bench.py switches on for a rank with ONE host worker): inside
    mprg_forest_level the LDS classes of the clustering loop go to the side stream, the general form for the problems no class holds
    runs beside them and once more after both.  Same trees as the per-step host's first forest."""
    import make_prg_amd.forest as forest
    monkeypatch.setattr(forest, "KM_SIDE_STREAMS", True)
    eng = ForestEngine(hip, 5, 7)
    eng.load([load_alignment_text(t) for t in texts])
    eng.run_forest()
    first = dump(eng, len(texts))
    for _ in range(2):
        reset(eng)
        eng.run_forest()
        assert eng.counters["syncs"] == 1 and eng.counters.get("plan_misses", 0) == 0
        assert dump(eng, len(texts)) == first


@pytest.mark.parametrize("step,col", [(0, 1), (1, 1), (1, 3), (3, 0), (4, 1), (5, 2)])
def test_small_capacities_fall_back(hip, texts, step, col):
    eng = ForestEngine(hip, 5, 7)
    eng.load([load_alignment_text(t) for t in texts[:60]])
    eng.run_forest()
    first = dump(eng, 60)
    lv = next(l for l in eng._plan["levels"] if l[step][col] > 0)
    lv[step][col] -= 1
    reset(eng)
    eng.run_forest()
    assert eng.counters.get("plan_misses", 0) == 1 and dump(eng, 60) == first
    reset(eng)
    eng.run_forest()
    assert eng.counters.get("plan_misses", 0) == 0 and eng.counters["syncs"] == 1 and dump(eng, 60) == first


def test_first_seen_batch_is_sized_from_another_batch(hip, texts, monkeypatch):
    """Capacities predicted from ANOTHER batch's totals (what the command line's chunks and a rank's shard get): same trees as the
    per-step host, one wait per forest; with far too little room, levels are enqueued again / left to the per-step host."""
    from make_prg_amd import forest
    a_texts, b_texts = texts[0::2], texts[1::2]
    a = ForestEngine(hip, 5, 7)
    a.load([load_alignment_text(t) for t in a_texts])
    a.run_forest()
    donor = a.plan_export()
    b0 = ForestEngine(hip, 5, 7)
    b0.load([load_alignment_text(t) for t in b_texts])
    b0.plan_donor = None
    hip.plan_donor = None                      # (a's forest left its totals on the backend: this one must run the per-step host)
    b0.run_forest()
    waits_exact = b0.counters["syncs"]
    want = dump(b0, len(b_texts))
    for head, floor, retries in ((forest.PLAN_HEAD, forest.PLAN_FLOOR, forest.SPEC_RETRIES), (0.3, 0.0, 6), (0.3, 0.0, 1)):
        monkeypatch.setattr(forest, "PLAN_HEAD", head)
        monkeypatch.setattr(forest, "PLAN_FLOOR", floor)
        monkeypatch.setattr(forest, "PLAN_SPREAD", forest.PLAN_SPREAD if head >= 1 else 0.0)
        monkeypatch.setattr(forest, "SPEC_RETRIES", retries)
        b = ForestEngine(hip, 5, 7)
        b.load([load_alignment_text(t) for t in b_texts])
        b.plan_donor = donor
        b.run_forest()
        waits, misses = b.counters["syncs"], b.counters["plan_misses"]
        assert dump(b, len(b_texts)) == want
        if head >= 1:
            assert misses == 0 and waits == 1 and waits_exact > 20
        else:
            assert misses >= 1 and b.counters["plan_resumes"] == (1 if misses > retries else 0)

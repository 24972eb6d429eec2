"""Levels without host waits (mprg_forest_level, forest._forest_speculative): the second forest of a resident batch is enqueued
from the first one's totals, the device counts its own items and flags a total that does not fit.  Same trees, same PRGs; one
wait per forest; a plan that does not fit falls back to the per-step host.  (Emulation backend: the same kernel sources on the CPU.)"""
import numpy as np
import pytest

from make_prg_amd.forest import ForestEngine
from make_prg_amd.msa import load_alignment_text
from make_prg_amd.utils.synthetic import synth_config_fasta
from tests.emu.backend import EmuBackend
from tests.random_msas import random_cases


def reset(eng):
    for k in eng.counters:
        eng.counters[k] = 0 if k != "arena_bytes" else eng.counters[k]


def dump(eng, n):
    prgs = eng.assemble_prgs(want_index=True)
    # (where a level's nodes sit in the node table depends on the order in which workgroups took their places — atomics — on the GPU:
    #  what is compared is independent of it: text, node count, how many nodes of each kind, the PRG index with its preorder node ids)
    return prgs, eng.n_nodes, np.bincount(eng.tab["kind"], minlength=3).tolist(), [eng.prg_index(i) for i in range(n) if prgs[i] is not None]


@pytest.fixture(scope="module")
def batch(golden_integration):
    texts = [synth_config_fasta("B", s) for s in range(4)] + random_cases(5, 24)
    texts += [l["fasta"] for c in golden_integration["cases"] if c["case"] in ("fails_2", "contains_n", "nested_snps_seq_backgrounds") and
              (c["N"], c["L"]) == (5, 7) for l in c["loci"]]
    return texts


def test_second_forest_runs_without_waits_and_is_identical(batch):
    eng = ForestEngine(EmuBackend(), 5, 7)
    eng.load([load_alignment_text(t) for t in batch])
    eng.run_forest()
    first = dump(eng, len(batch))
    waits_exact = eng.counters["syncs"]
    assert any(p is None for p in first[0]), "the batch holds a locus the curation policy drops"
    for _ in range(2):
        reset(eng)
        eng.run_forest()
        assert eng.counters["syncs"] == 1 and eng.counters.get("plan_misses", 0) == 0 and waits_exact > 10
        again = dump(eng, len(batch))
        assert again == first
        assert [type(e) for e in eng.errors.values()] and set(eng.errors) == {i for i, p in enumerate(first[0]) if p is None}


@pytest.mark.parametrize("step,col", [(0, 0), (0, 1), (1, 1), (1, 3), (2, 0), (3, 0), (4, 0), (4, 1), (5, 0), (5, 2)])
def test_a_capacity_that_is_too_small_falls_back(batch, step, col):
    """Every kind of total one short of what the level needs: the device flags it before anything is written past a buffer, the
    host repeats the forest with exact sizes — and plans again."""
    eng = ForestEngine(EmuBackend(), 5, 7)
    eng.load([load_alignment_text(t) for t in batch])
    eng.run_forest()
    first = dump(eng, len(batch))
    lv = next(l for l in eng._plan["levels"] if l[step][col] > 0)
    lv[step][col] -= 1
    reset(eng)
    eng.run_forest()
    assert eng.counters.get("plan_misses", 0) == 1
    assert dump(eng, len(batch)) == first
    reset(eng)
    eng.run_forest()
    assert eng.counters.get("plan_misses", 0) == 0 and eng.counters["syncs"] == 1 and dump(eng, len(batch)) == first


def test_a_forest_deeper_or_wider_than_the_plan_falls_back(batch):
    eng = ForestEngine(EmuBackend(), 5, 7)
    eng.load([load_alignment_text(t) for t in batch])
    eng.run_forest()
    first = dump(eng, len(batch))
    eng._plan["levels"].pop()                      # one level short
    reset(eng)
    eng.run_forest()
    assert eng.counters.get("plan_misses", 0) == 1 and dump(eng, len(batch)) == first
    # node table / row pool smaller than the forest needs (their capacities are otherwise the previous forest's sizes plus headroom)
    for key in ("n_nodes", "pool_used"):
        reset(eng)
        eng.run_forest()
        assert eng.counters.get("plan_misses", 0) == 0
        need = eng._plan[key]
        eng._plan[key] = need // 2
        eng._nodes_hint = eng._pool_hint = 0
        eng.cap_floor = 1                           # (test hook: no minimum table size)
        reset(eng)
        eng.run_forest()
        assert eng.counters.get("plan_misses", 0) == (1 if need >= 2 else 0), key
        assert dump(eng, len(batch)) == first


# ---- a batch seen for the FIRST time: capacities predicted from ANOTHER batch's totals (forest._caps_predicted) ---------------------
@pytest.fixture(scope="module")
def other_batch(golden_integration):
    texts = [synth_config_fasta("B", s) for s in range(20, 23)] + random_cases(11, 30)
    texts += [l["fasta"] for c in golden_integration["cases"] if c["case"] in ("contains_n", "nested_snps_seq_backgrounds") and
              (c["N"], c["L"]) == (5, 7) for l in c["loci"]]
    return texts


def _exact(texts):
    eng = ForestEngine(EmuBackend(), 5, 7)
    eng.load([load_alignment_text(t) for t in texts])
    eng.run_forest()
    eng.forest_waits = eng.counters["syncs"]
    return eng, dump(eng, len(texts))


def _first_seen(texts, donor):
    eng = ForestEngine(EmuBackend(), 5, 7)
    eng.load([load_alignment_text(t) for t in texts])
    eng.plan_donor = donor
    eng.run_forest()
    eng.forest_waits = eng.counters["syncs"]
    return eng, dump(eng, len(texts))


def test_first_forest_sized_from_another_batch(batch, other_batch):
    a, _ = _exact(batch)
    donor = a.plan_export()
    assert donor is not None and donor["n_roots"] == len(a.ok)
    want_eng, want = _exact(other_batch)
    eng, got = _first_seen(other_batch, donor)
    assert got == want
    misses = eng.counters.get("plan_misses", 0)
    assert eng.forest_waits == 1 + misses and eng.forest_waits < want_eng.forest_waits // 4
    assert eng.counters.get("plan_resumes", 0) == 0
    # ... and the other way round, and a batch from its own export
    assert _first_seen(batch, want_eng.plan_export())[1] == dump(a, len(batch))
    assert _first_seen(batch, donor)[1] == dump(a, len(batch))


@pytest.mark.parametrize("head,floor,retries", [(0.4, 0.0, 6), (0.05, 0.0, 6), (0.4, 0.0, 0), (0.9, 1.0, 2)])
def test_a_prediction_that_is_too_small_costs_levels_not_the_forest(batch, other_batch, monkeypatch, head, floor, retries):
    """Capacities far below what the batch needs: every level that does not fit is enqueued again with more room (the levels before
    it are kept), and after `retries` such rounds the per-step host takes over AT that level.  Same trees either way."""
    from make_prg_amd import forest
    a, _ = _exact(batch)
    _, want = _exact(other_batch)
    monkeypatch.setattr(forest, "PLAN_HEAD", head)
    monkeypatch.setattr(forest, "PLAN_SPREAD", 0.0)
    monkeypatch.setattr(forest, "PLAN_FLOOR", floor)
    monkeypatch.setattr(forest, "SPEC_RETRIES", retries)
    eng, got = _first_seen(other_batch, a.plan_export())
    assert got == want
    assert eng.counters["plan_misses"] >= 1
    assert eng.counters.get("plan_resumes", 0) == (1 if eng.counters["plan_misses"] > retries else 0)
    # the forest left a plan of its own: the next one of the same batch needs no second look
    reset(eng)
    eng.run_forest()
    assert eng.counters["syncs"] == 1 and eng.counters.get("plan_misses", 0) == 0 and dump(eng, len(other_batch)) == want


def test_a_level_with_a_big_problem_is_left_to_the_per_step_host(batch, other_batch, monkeypatch):
    """MPRG_CAP_BIG: the device-counted path stops at a level that holds a clustering problem whose count matrix is BIG (the per-step
    host has wider launch forms for it) and the per-step host goes on from there."""
    from make_prg_amd import forest
    a, _ = _exact(batch)
    _, want = _exact(other_batch)
    monkeypatch.setattr(forest, "KM_BIG_BYTES", 2048)
    eng, got = _first_seen(other_batch, a.plan_export())
    assert got == want
    assert eng.counters.get("plan_resumes", 0) == 1 and eng.counters["plan_misses"] == 1 and eng._plan is None

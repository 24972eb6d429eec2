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

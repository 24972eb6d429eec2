"""Seeded random small alignments: both hosts (array-at-a-time forest, node objects) on the emulation backend
must agree with the oracle on PRG, recursion tree and prg_index, for several (max_nesting, min_match_length)."""
import pytest

from tests import parity_common as pc
from tests.emu.backend import EmuBackend
from tests.random_msas import random_cases


@pytest.fixture(scope="module")
def emu():
    return EmuBackend()


@pytest.mark.parametrize("N,L,seed", [(5, 7, 11), (5, 3, 12), (2, 1, 13), (1, 7, 14), (5, 2, 15)])
def test_forest_host(emu, N, L, seed, monkeypatch):
    monkeypatch.setattr(pc, "ENGINE", "forest")
    pc.check_vs_oracle(emu, random_cases(seed, 150), N, L)


@pytest.mark.parametrize("N,L,seed", [(5, 7, 21), (3, 3, 22)])
def test_node_host(emu, N, L, seed, monkeypatch):
    monkeypatch.setattr(pc, "ENGINE", "nodes")
    pc.check_vs_oracle(emu, random_cases(seed, 80), N, L)

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


def test_speculative_k_rounds_give_the_same_trees(emu, golden_synthetic, monkeypatch):
    """forest.ForestEngine.k_slots > 1 fits several k per launch and replays the reference's decisions afterwards."""
    from make_prg_amd.forest import ForestEngine
    monkeypatch.setattr(ForestEngine, "k_slots", 3)
    assert pc.check_synthetic(emu, golden_synthetic, configs=("B",), limit=16) == 16
    assert pc.check_synthetic(emu, golden_synthetic, configs=("C",), limit=3) == 3

"""Deterministic synthetic MSAs for the BASELINE.json configs (generator spec: SURVEY.md §8(d)).

One NumPy `default_rng(seed)` stream per MSA; the call order below is part of the spec (fixtures depend on it).
"""
from typing import List, Tuple

import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(rng, x, p):
    y = x.copy()
    m = rng.random(x.shape[0]) < p
    y[m] = rng.integers(0, 4, int(m.sum()))
    return y


def synth_rows(seed: int, S: int, C: int, n_clades: int) -> List[bytes]:
    """Rows of one synthetic MSA (bytes over ACGT-), ids are s0..s{S-1}."""
    rng = np.random.default_rng(seed)
    root = rng.integers(0, 4, C)
    clades = [_mutate(rng, root, 0.03) for _ in range(n_clades)]
    rows = []
    for _ in range(S):
        row = _mutate(rng, clades[int(rng.integers(0, n_clades))], 0.004)
        txt = _BASES[row].copy()
        for st in np.nonzero(rng.random(C) < 0.0005)[0]:
            ln = int(rng.integers(1, 12))
            txt[st:st + ln] = ord("-")
        rows.append(txt.tobytes())
    return rows


def config_shape(config: str, seed: int) -> Tuple[int, int, int]:
    """(S, C, n_clades) for a BASELINE config letter: B = 1k x (50x500), C = 30k pan-genome, D = deep single MSA."""
    if config == "B":
        return 50, 500, 2 + seed % 5
    if config == "C":
        g = np.random.default_rng([seed, 1])
        S = int(np.clip(round(g.normal(100, 20)), 20, 300))
        C = int(g.integers(1000, 3001))
        return S, C, 2 + seed % 5
    if config == "D":
        return 10000, 20000, 8
    raise ValueError(config)


def synth_fasta(seed: int, S: int, C: int, n_clades: int) -> str:
    rows = synth_rows(seed, S, C, n_clades)
    return "".join(f">s{i}\n{r.decode()}\n" for i, r in enumerate(rows))


def synth_config_fasta(config: str, seed: int) -> str:
    return synth_fasta(seed, *config_shape(config, seed))

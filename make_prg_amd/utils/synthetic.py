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


def synth_rows_deep(seed: int, S: int, C: int, fanout=(3, 3, 3, 3, 3), rates=(0.6, 0.5, 0.4, 0.35, 0.3),
                    window: int = 60, period: int = 200, p_background: float = 0.0005, p_row: float = 0.0002,
                    p_gap: float = 0.0001) -> List[bytes]:
    """Rows of a HIERARCHICAL alignment (ids s0..): a conserved backbone with one hyper-variable window of `window`
    columns every `period` columns.  A tree of clades (fanout[l] children per node at level l) is grown from a random
    root: inside the windows a child differs from its parent with substitution rate rates[l], elsewhere with
    p_background; the S rows are drawn from the leaf clades with rare private substitutions (p_row) and gap runs
    (p_gap).  Unlike synth_rows (one flat layer of clades: at 10k x 20k a root and one multi-allele leaf) the recursion
    NESTS here: every window is a non-match interval whose rows KMeans separates by top clade, the children separate the
    sub-clades, and so on down to max_nesting — sibling clades stay further apart than the one-reference-like threshold
    (20 % of the width) at every level.  Used for the parity-checked deep case "Ddeep" (2 000 x 4 000, -N 7); the call
    order on default_rng(seed) is part of the fixture."""
    rng = np.random.default_rng(seed)
    in_window = (np.arange(C) % period) >= (period - window)
    level = [rng.integers(0, 4, C)]
    for fan, rate in zip(fanout, rates):
        pvec = np.where(in_window, rate, p_background)
        nxt = []
        for parent in level:
            for _ in range(fan):
                child = parent.copy()
                m = rng.random(C) < pvec
                child[m] = rng.integers(0, 4, int(m.sum()))
                nxt.append(child)
        level = nxt
    rows = []
    for _ in range(S):
        row = _mutate(rng, level[int(rng.integers(0, len(level)))], p_row)
        txt = _BASES[row].copy()
        for st in np.nonzero(rng.random(C) < p_gap)[0]:
            ln = int(rng.integers(1, 12))
            txt[st:st + ln] = ord("-")
        rows.append(txt.tobytes())
    return rows


def synth_deep_fasta(seed: int, S: int, C: int) -> str:
    return "".join(f">s{i}\n{r.decode()}\n" for i, r in enumerate(synth_rows_deep(seed, S, C)))

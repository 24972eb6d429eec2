"""GPU-box helper: the seeded generator of small nasty alignments (tests/random_msas.py: gaps, all-gap rows/columns,
ambiguity codes, N, lower case, duplicate rows, disallowed bases) at scale, several (max_nesting, min_match_length),
HIP path vs oracle: PRG, node count, or the same SequenceCurationError.   usage: parity_sweep_nasty.py [cases per combo]"""
import multiprocessing as mp
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

COMBOS = [(5, 7), (5, 3), (2, 1), (1, 7), (5, 2), (7, 4)]


def _one(args):
    text, N, L = args
    import oracle.from_msa_oracle as orc
    try:
        prg, b, root = orc.build_locus_from_text(text, N, L)
        return prg, b.next_node_id
    except orc.SequenceCurationError:
        return "SequenceCurationError", -1


def medium_cases(seed, n):
    """Larger relatives of tests/random_msas.py: 5-60 rows x 30-400 columns, 1-6 clades, indels, sparse N / ambiguity codes."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        S, C = int(rng.integers(5, 61)), int(rng.integers(30, 401))
        base = rng.integers(0, 4, C)
        clades = []
        for _c in range(int(rng.integers(1, 7))):
            y = base.copy()
            m = rng.random(C) < rng.choice([0.01, 0.03, 0.1])
            y[m] = rng.integers(0, 4, int(m.sum()))
            clades.append(y)
        rows = []
        for i in range(S):
            y = clades[int(rng.integers(0, len(clades)))].copy()
            m = rng.random(C) < rng.choice([0.0, 0.004, 0.02])
            y[m] = rng.integers(0, 4, int(m.sum()))
            txt = np.frombuffer(b"ACGT", np.uint8)[y].copy()
            for st in np.nonzero(rng.random(C) < 0.004)[0]:
                txt[st:st + int(rng.integers(1, 12))] = ord("-")
            if rng.random() < 0.05:
                for st in np.nonzero(rng.random(C) < 0.01)[0]:
                    txt[st] = ord("RYKMSWN"[int(rng.integers(0, 7))])
            rows.append(f">r{i}\n{txt.tobytes().decode()}\n")
        out.append("".join(rows))
    return out


from make_prg_amd.utils.misc import effective_cpus
N_PROCS = effective_cpus()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    from tests.random_msas import random_cases
    if len(sys.argv) > 2 and sys.argv[2] == "medium":
        random_cases = medium_cases
    import oracle.from_msa_oracle as orc
    orc.build_kmeans_lib()
    sets = {c: random_cases(1000 + i, n) for i, c in enumerate(COMBOS)}
    t0 = time.time()
    with mp.get_context("fork").Pool(N_PROCS) as pool:
        want = {c: pool.map(_one, [(t, c[0], c[1]) for t in sets[c]], chunksize=16) for c in COMBOS}
    print(f"oracle: {n * len(COMBOS)} alignments in {time.time() - t0:.0f}s", flush=True)
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.engine import SequenceCurationError
    from make_prg_amd.forest import ForestEngine
    from make_prg_amd.msa import load_alignment_text
    be = HipBackend(0)
    total_bad = 0
    for (N, L) in COMBOS:
        texts = sets[(N, L)]
        msas, keep = [], []
        pre_err = {}
        for i, t in enumerate(texts):
            try:
                msas.append(load_alignment_text(t)); keep.append(i)
            except ValueError as e:
                pre_err[i] = e
        eng = ForestEngine(be, N, L)
        eng.load(msas)
        eng.run_forest()
        prgs = eng.assemble_prgs()
        nodes = np.bincount(eng.tab["msa"], minlength=len(msas)) if len(eng.tab["msa"]) else np.zeros(len(msas), int)
        bad = []
        for j, i in enumerate(keep):
            w = want[(N, L)][i]
            if prgs[j] is None:
                ok = isinstance(eng.errors[j], SequenceCurationError) and w[0] == "SequenceCurationError"
            else:
                ok = (prgs[j], int(nodes[j])) == w
            if not ok:
                bad.append(i)
        print(f"N={N} L={L}: {len(keep)} alignments ({sum(1 for w in want[(N, L)] if w[1] < 0)} curation errors), mismatches {len(bad)} {bad[:5]}", flush=True)
        total_bad += len(bad)
    sys.exit(1 if total_bad else 0)

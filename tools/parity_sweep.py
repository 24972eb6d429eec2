"""GPU-box helper: a large parity sweep outside the test suite: N fresh config-C alignments (seeds from S0), every PRG and
node count of the HIP path against the oracle (worker processes are forked before the GPU is touched).
usage: parity_sweep.py [N] [S0]"""
import multiprocessing as mp
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def _one(seed):
    import oracle.from_msa_oracle as orc
    from make_prg_amd.utils.synthetic import synth_config_fasta
    prg, b, root = orc.build_locus_from_text(synth_config_fasta("C", seed), 5, 7)
    return prg, b.next_node_id


def _load(seed):
    from make_prg_amd.msa import load_alignment_text
    from make_prg_amd.utils.synthetic import synth_config_fasta
    return load_alignment_text(synth_config_fasta("C", seed))


from make_prg_amd.utils.misc import effective_cpus
N_PROCS = effective_cpus()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
    seeds = list(range(s0, s0 + n))
    import oracle.from_msa_oracle as orc
    orc.build_kmeans_lib()
    t0 = time.time()
    with mp.get_context("fork").Pool(N_PROCS) as pool:
        want = pool.map(_one, seeds, chunksize=2)
        msas = pool.map(_load, seeds, chunksize=8)
    print(f"oracle: {n} alignments in {time.time() - t0:.0f}s on {N_PROCS} processes", flush=True)
    from make_prg_amd.backend import HipBackend
    from make_prg_amd.forest import ForestEngine
    eng = ForestEngine(HipBackend(0), 5, 7)
    eng.load(msas)
    t0 = time.time()
    eng.run_forest()
    prgs = eng.assemble_prgs()
    dt = time.time() - t0
    nodes = np.bincount(eng.tab["msa"], minlength=n)
    bad = [s for s, g, (w, nn), k in zip(seeds, prgs, want, nodes) if g != w or k != nn]
    print(f"HIP path: {n} alignments in {dt:.2f}s; fits {int(eng.counters['fits'])}; mismatches: {len(bad)} {bad[:10]}")
    sys.exit(1 if bad else 0)

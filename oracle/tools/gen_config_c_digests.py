"""TEST INFRASTRUCTURE: digest fixture of BASELINE.json config C at FULL size (30 000 alignments, seeds 0..29999,
-N 5 -L 7), produced by the oracle (oracle/from_msa_oracle.py + kmeans_oracle.c, pinned to the real reference by
oracle/tools/gen_golden.py).  One 12-byte record per seed: sha256(PRG string)[:8] + uint32 node count (little endian).
The `-m gpu` test tests/test_gpu_config_c_full.py builds all 30 000 on the MI355X and compares every record; bench.py
checks the records of the seeds it times.  Runs wherever the oracle runs (pure CPU); ~8 min on 256 threads, hours on 8.

usage: python oracle/tools/gen_config_c_digests.py [--out tests/golden/config_c_digests.bin] [--n 30000] [--procs P]"""
import argparse
import hashlib
import multiprocessing as mp
import os
import struct
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def digest_record(prg: str, n_nodes: int) -> bytes:
    return hashlib.sha256(prg.encode()).digest()[:8] + struct.pack("<I", n_nodes)


def _one(seed):
    import oracle.from_msa_oracle as orc
    from make_prg_amd.utils.synthetic import synth_config_fasta
    prg, b, root = orc.build_locus_from_text(synth_config_fasta("C", seed), 5, 7)
    return digest_record(prg, b.next_node_id)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join("tests", "golden", "config_c_digests.bin"))
    ap.add_argument("--n", type=int, default=30000)
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    a = ap.parse_args()
    import oracle.from_msa_oracle as orc
    orc.build_kmeans_lib()
    t0 = time.time()
    with mp.get_context("fork").Pool(a.procs) as pool:
        recs = pool.map(_one, range(a.n), chunksize=4)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "wb") as fh:
        fh.write(b"".join(recs))
    print(f"{a.n} records -> {a.out} in {time.time() - t0:.0f}s on {a.procs} processes")

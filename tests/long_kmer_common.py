"""k-mer sizes above 16 (verified hash keys in the k-mer dictionary): the real reference's answers for -L 17 / 20 / 24 / 33
(tests/golden/long_kmer.json.gz, oracle/tools/gen_long_kmer_golden.py)."""
from tests import parity_common as pc
from tests.conftest import load_golden
from make_prg_amd.utils.synthetic import synth_rows_deep


def check_long_kmers(backend):
    g = load_golden("long_kmer.json.gz")
    n = 0
    for r in g["loci"]:
        rows = synth_rows_deep(r["seed"], r["S"], r["C"], fanout=(3, 3, 2), rates=(0.6, 0.5, 0.4), window=r["window"], period=r["period"])
        text = "".join(f">s{i}\n{x.decode()}\n" for i, x in enumerate(rows))
        assert pc.sha(text) == r["fasta_sha256"], "the generator changed under the fixture"
        for engine in ("forest", "nodes"):
            got, eng = pc.run_batch(backend, [text], r["N"], r["L"], engine=engine)
            pc.check_against_expect(got[0], r["expect"], f"long k-mer L={r['L']} ({engine})")
            assert int(eng.counters["fits"]) > 0
        n += 1
    return n

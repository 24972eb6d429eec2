"""The oracle (oracle/) against the committed golden vectors produced by the real reference
(oracle/tools/gen_golden.py; pinned configuration recorded in each fixture's `meta`)."""
import hashlib
import json

import numpy as np
import pytest

import oracle.from_msa_oracle as orc
from make_prg_amd.utils.synthetic import synth_fasta


def sha(obj):
    if isinstance(obj, str):
        obj = obj.encode()
    elif not isinstance(obj, (bytes, bytearray)):
        obj = json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()
    return hashlib.sha256(obj).hexdigest()


def check_locus(text, N, L, expect):
    if "error" in expect:
        with pytest.raises(orc.SequenceCurationError):
            orc.build_locus_from_text(text, N, L)
        return
    prg, b, root = orc.build_locus_from_text(text, N, L)
    assert prg == expect["prg"]
    assert sha(orc.encode_prg_bytes(prg)) == expect["bin_sha256"]
    assert sha(orc.gfa_text(prg)) == expect["gfa_sha256"]
    assert sha(orc.tree_dump(root)) == expect["tree_sha256"]
    assert sorted([s, e, n] for (s, e), n in b.prg_index.items()) == expect["prg_index"]
    assert (b.next_node_id, b.site_num) == (expect["next_node_id"], expect["site_num"])
    if "tree" in expect:
        assert orc.tree_dump(root) == expect["tree"]
        assert orc.gfa_text(prg) == expect["gfa"]
        assert orc.encode_prg_bytes(prg).hex() == expect["bin_hex"]


def test_meta_records_pinned_configuration(golden_integration):
    m = golden_integration["meta"]
    assert m["n_init"] == 10 and m["OMP_NUM_THREADS"] == "1" and m["OPENBLAS_CORETYPE"] == "Haswell"


def test_integration_cases(golden_integration):
    n = 0
    for case in golden_integration["cases"]:
        for locus in case["loci"]:
            check_locus(locus["fasta"], case["N"], case["L"], locus["expect"])
            n += 1
    assert n >= 30


def test_function_level_traces(golden_integration):
    ncons = nclu = 0
    for case in golden_integration["cases"]:
        for locus in case["loci"]:
            for call in locus.get("calls", []):
                rows = call["rows"]
                assert orc.consensus_string(rows) == call["consensus"]
                _, _, allv = orc.partition_intervals(call["consensus"], call["L"], rows)
                assert [list(x) for x in allv] == call["intervals"]
                ncons += 1
            for call in locus.get("cluster_calls", []):
                aln = [(i, "", r) for i, r in zip(call["ids"], call["rows"])]
                res = orc.cluster_rows(aln, call["k"])
                assert res.clustered_ids == call["clustered_ids"]
                assert res.sequences == call["sequences"]
                nclu += 1
    assert ncons > 100 and nclu > 10


def test_synthetic_loci(golden_synthetic):
    for rec in golden_synthetic["loci"]:
        text = synth_fasta(rec["seed"], rec["S"], rec["C"], rec["n_clades"])
        assert sha(text) == rec["fasta_sha256"], "synthetic generator drifted"
        check_locus(text, rec["N"], rec["L"], rec["expect"])


def test_kmeans_known_answers(golden_kmeans):
    assert len(golden_kmeans["fits"]) >= 100
    for fit in golden_kmeans["fits"]:
        D, V = fit["shape"]
        X = np.frombuffer(bytes.fromhex(fit["counts_i16_hex"]), dtype="<i2").reshape(D, V).astype(np.float64)
        labels, dbg = orc.kmeans_fit_predict(X, fit["k"], want_debug=True)
        assert labels.tolist() == fit["labels"]
        assert dbg["fit_labels"].tolist() == fit["fit_labels"]
        assert dbg["pp"].tolist() == fit["pp"]
        assert float(dbg["inertia"]).hex() == fit["inertia"]
        assert dbg["n_iter"] == fit["n_iter"]


def test_random_stream_matches_numpy_randomstate():
    import ctypes
    lib = orc._lib()
    out = np.zeros(1000)
    lib.mprg_oracle_random_sample(ctypes.c_uint32(2), 1000, ctypes.c_void_p(out.ctypes.data))
    assert np.array_equal(out, np.random.RandomState(2).random_sample(1000))


def test_load_time_consensus_array_form_equals_the_column_by_column_form():
    """A0 (utils/seq_utils.py:246-290): the array-at-a-time majority consensus draws the same bases from the same
    random.Random stream as the plain column-by-column restatement, on random matrices with gaps, N, ambiguity codes."""
    import numpy as np
    from make_prg_amd import msa
    rng = np.random.default_rng(1)
    alpha = np.frombuffer(b"ACGT-NRYK", np.uint8)
    for trial in range(200):
        S, C = int(rng.integers(1, 12)), int(rng.integers(1, 40))
        p = rng.random(len(alpha))
        m = alpha[rng.choice(len(alpha), size=(S, C), p=p / p.sum())].astype(np.uint8)
        upto = None if trial % 3 else int(rng.integers(0, C + 1))
        assert (msa._majority_consensus(m, upto) == msa._majority_consensus_by_column(m, upto)).all()

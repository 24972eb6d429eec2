"""Shared parity checks: the engine (on whatever backend the caller passes) against golden vectors / the oracle."""
import hashlib
import json

import numpy as np

import oracle.from_msa_oracle as orc
from make_prg_amd.engine import BatchEngine, SequenceCurationError, build_prg, tree_dump
from make_prg_amd.msa import load_alignment_text
from make_prg_amd.utils.gfa import GFA_Output
from make_prg_amd.utils.prg_encoder import PrgEncoder
from make_prg_amd.utils.synthetic import synth_fasta


def product_bin_bytes(prg: str) -> bytes:
    import io
    enc = PrgEncoder()
    buf = io.BytesIO()
    enc.write(enc.encode(prg), buf)
    return buf.getvalue()


def sha(obj):
    if isinstance(obj, str):
        obj = obj.encode()
    elif not isinstance(obj, (bytes, bytearray)):
        obj = json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()
    return hashlib.sha256(obj).hexdigest()


ENGINE = "forest"          # which host the parity checks drive: "forest" (array-at-a-time) or "nodes" (engine.py)


def run_batch(backend, texts, N, L, engine=None):
    msas = [load_alignment_text(t, defer_n=True) for t in texts]     # N columns: counted on the device
    if (engine or ENGINE) == "nodes":
        eng = BatchEngine(backend, N, L)
        res = eng.build(msas)
        out = []
        for r, m in zip(res, msas):
            if r.error is not None:
                out.append(dict(error=type(r.error).__name__))
                continue
            prg, index, site = build_prg(eng, r)
            tree = tree_dump(eng, r, m.ids)
            out.append(dict(prg=prg, tree=tree, site_num=site, next_node_id=len(tree),
                            prg_index=sorted([s, e, r.nodes[ni].node_id] for (s, e), ni in index.items())))
        return out, eng
    from make_prg_amd.forest import ForestEngine
    eng = ForestEngine(backend, N, L)
    eng.load(msas)
    eng.run_forest()
    prgs = eng.assemble_prgs(want_index=True)
    out = []
    for i, (p, m) in enumerate(zip(prgs, msas)):
        if p is None:
            out.append(dict(error=type(eng.errors[i]).__name__))
            continue
        tree = eng.tree_dump(i, m.ids)
        out.append(dict(prg=p, tree=tree, site_num=5 + 2 * int(eng.site_count[i]), next_node_id=len(tree),
                        prg_index=eng.prg_index(i)))
    return out, eng


def check_against_expect(got, expect, tag=""):
    if "error" in expect:
        assert got.get("error") == expect["error"], tag
        return
    assert "error" not in got, f"{tag}: unexpected {got.get('error')}"
    assert got["prg"] == expect["prg"], f"{tag}: PRG differs"
    # the PRODUCT's encoders (what the CLI writes into .bin / .gfa), not the oracle's
    assert sha(product_bin_bytes(got["prg"])) == expect["bin_sha256"], tag
    assert sha(GFA_Output.gfa_text(got["prg"])) == expect["gfa_sha256"], tag
    assert sha(got["tree"]) == expect["tree_sha256"], f"{tag}: recursion tree differs"
    assert got["prg_index"] == expect["prg_index"], tag
    assert (got["next_node_id"], got["site_num"]) == (expect["next_node_id"], expect["site_num"]), tag


def check_integration(backend, golden, batch_all=True):
    """Every integration case of the reference's own test-suite; loci of one case go through ONE batched build."""
    n = 0
    for case in golden["cases"]:
        texts = [l["fasta"] for l in case["loci"]]
        got, _ = run_batch(backend, texts, case["N"], case["L"])
        for g, l in zip(got, case["loci"]):
            check_against_expect(g, l["expect"], f"{case['case']}/{l['locus']}")
            n += 1
    return n


def check_synthetic(backend, golden, configs=("B", "C", "Dsmall"), limit=None):
    recs = [r for r in golden["loci"] if r["config"] in configs][:limit]
    by_params = {}
    for r in recs:
        by_params.setdefault((r["N"], r["L"]), []).append(r)
    n = 0
    for (N, L), rs in by_params.items():
        texts = [synth_fasta(r["seed"], r["S"], r["C"], r["n_clades"]) for r in rs]
        got, eng = run_batch(backend, texts, N, L)          # one batch: level-synchronous over all loci
        for g, r in zip(got, rs):
            check_against_expect(g, r["expect"], f"synthetic {r['config']}{r['seed']}")
            n += 1
    return n


def check_vs_oracle(backend, texts, N=5, L=7):
    """Engine vs oracle on arbitrary inputs (used for seeded random and edge cases)."""
    got, eng = run_batch(backend, texts, N, L)
    for g, t in zip(got, texts):
        try:
            prg, b, root = orc.build_locus_from_text(t, N, L)
        except orc.SequenceCurationError:
            assert g.get("error") == "SequenceCurationError"
            continue
        assert "error" not in g, g
        assert g["prg"] == prg
        assert g["tree"] == orc.tree_dump(root)
        assert g["prg_index"] == sorted([s, e, n] for (s, e), n in b.prg_index.items())
    return eng


def check_compact_columns(backend) -> int:
    """mprg_compact_columns (A8) behind remove_columns_full_of_gaps_from_MSA against NumPy: narrow, wide (places in LDS) and
    very wide (places recomputed per tile) alignments, all-gap runs at the edges, nothing to drop, everything dropped."""
    from make_prg_amd import device
    from make_prg_amd.msa import MSA
    from make_prg_amd.utils import seq_utils as su
    rng = np.random.default_rng(11)
    device.set_backend(backend)
    n = 0
    try:
        for rows, cols, p_gapcol in ((3, 17, 0.3), (70, 700, 0.5), (130, 3000, 0.1), (5, 9000, 0.4), (4, 50, 0.0), (4, 50, 1.0)):
            data = rng.choice(np.frombuffer(b"ACGT-", np.uint8), size=(rows, cols), p=[0.22, 0.22, 0.22, 0.22, 0.12])
            drop = rng.random(cols) < p_gapcol
            data[:, drop] = ord("-")
            data[:, :2] = ord("-")
            msa = MSA.from_strings([r.tobytes().decode() for r in data])
            got = su.remove_columns_full_of_gaps_from_MSA(msa)
            want = data[:, ~(data == ord("-")).all(axis=0)]
            assert got.get_alignment_length() == want.shape[1]
            assert np.array_equal(got.data, want)
            n += 1
    finally:
        device.set_backend(None)
    return n

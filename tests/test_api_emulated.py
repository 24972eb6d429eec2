"""The reference-shaped Python API (PrgBuilder, NodeFactory, get_consensus_from_MSA, IntervalPartitioner,
kmeans_cluster_seqs, encoders) on the emulation backend against the reference's traced calls and truths."""
import hashlib
import io
import pickle

import pytest

from make_prg_amd import device
from make_prg_amd.msa import MSA
from tests.emu.backend import EmuBackend


@pytest.fixture(autouse=True, scope="module")
def _emu():
    device.set_backend(EmuBackend())
    yield
    device.set_backend(None)


def test_function_level_traces(golden_integration):
    from make_prg_amd.from_msa.cluster_sequences import kmeans_cluster_seqs
    from make_prg_amd.from_msa.interval_partition import IntervalPartitioner
    from make_prg_amd.utils.seq_utils import get_consensus_from_MSA, remove_columns_full_of_gaps_from_MSA
    nc = ncl = 0
    for case in golden_integration["cases"]:
        for l in case["loci"]:
            for call in l.get("calls", []):
                msa = MSA.from_strings(call["rows"])
                assert get_consensus_from_MSA(msa) == call["consensus"]
                _, _, allv = IntervalPartitioner(call["consensus"], call["L"], msa).get_intervals()
                assert [[iv.start, iv.stop, "M" if iv.type.name == "Match" else "N"] for iv in allv] == call["intervals"]
                kept = remove_columns_full_of_gaps_from_MSA(msa)
                want = [c for c in range(len(call["rows"][0])) if any(r[c] != "-" for r in call["rows"])]
                assert kept.rows_as_strings() == ["".join(r[c] for c in want) for r in call["rows"]]
                nc += 1
            for call in l.get("cluster_calls", []):
                r = kmeans_cluster_seqs(MSA.from_strings(call["rows"], call["ids"]), call["k"])
                assert r.clustered_ids == call["clustered_ids"] and r.sequences == call["sequences"]
                ncl += 1
    assert nc > 100 and ncl > 30


def test_partition_of_bare_consensus_strings():
    """The reference's unit tests partition consensus strings without an alignment (tests/from_msa/test_interval_partition.py)."""
    from make_prg_amd.from_msa.interval_partition import IntervalPartitioner, IntervalType
    empty = MSA([])
    m, n, a = IntervalPartitioner("AAAAAAA*****", 7, empty).get_intervals()
    assert [(i.start, i.stop) for i in m] == [(0, 6)] and [(i.start, i.stop) for i in n] == [(7, 11)]
    m, n, a = IntervalPartitioner("TTATT", 7, empty).get_intervals()          # shorter than min_match_length: match
    assert [(i.start, i.stop, i.type) for i in a] == [(0, 4, IntervalType.Match)]
    m, n, a = IntervalPartitioner("**", 7, empty).get_intervals()
    assert [(i.start, i.stop, i.type) for i in a] == [(0, 1, IntervalType.NonMatch)]
    m, n, a = IntervalPartitioner("AAAAAAA**AA**AAAAAAA", 7, empty).get_intervals()   # short match absorbed
    assert [(i.start, i.stop) for i in a] == [(0, 6), (7, 12), (13, 19)]
    m, n, a = IntervalPartitioner("", 7, empty).get_intervals()
    assert a == []


def test_prg_builder_objects(golden_integration, tmp_path):
    from make_prg_amd.prg_builder import PrgBuilder
    from make_prg_amd.utils.gfa import GFA_Output
    from make_prg_amd.utils.prg_encoder import PrgEncoder
    n = 0
    for case in golden_integration["cases"]:
        for l in case["loci"]:
            e = l["expect"]
            path = tmp_path / "x.fa"
            path.write_text(l["fasta"])
            if "error" in e:
                with pytest.raises(Exception) as ei:
                    PrgBuilder(l["locus"], path, "fasta", case["N"], case["L"])
                assert type(ei.value).__name__ == e["error"]
                continue
            b = PrgBuilder(l["locus"], path, "fasta", case["N"], case["L"])
            prg = b.build_prg()
            assert prg == e["prg"]
            assert sorted([s, t, nd.node_id] for (s, t), nd in b.prg_index.items()) == e["prg_index"]
            assert (b.next_node_id, b.site_num) == (e["next_node_id"], e["site_num"])
            assert hashlib.sha256(GFA_Output.gfa_text(prg).encode()).hexdigest() == e["gfa_sha256"]
            buf = io.BytesIO()
            enc = PrgEncoder()
            enc.write(enc.encode(prg), buf)
            assert hashlib.sha256(buf.getvalue()).hexdigest() == e["bin_sha256"]
            assert pickle.loads(pickle.dumps(b, protocol=4)) == b
            assert b.get_node_given_interval(tuple(e["prg_index"][0][:2])).node_id == e["prg_index"][0][2]
            n += 1
    assert n >= 30

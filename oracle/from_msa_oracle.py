"""
oracle/from_msa_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement of the reference's `from_msa` hot path (make_prg v0.5.0) on plain Python data: an alignment is a
list of (id, description, row-string) triples; nodes are small dict-like objects.  It exists so that the HIP path
can be checked on any box (the Python reference cannot travel to the GPU box).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity status: PINNED.  `oracle/tools/gen_golden.py` runs the real reference (imported from /root/reference under
oracle/refshim, scikit-learn KMeans forced to n_init=10, OMP_NUM_THREADS=1, OPENBLAS_CORETYPE=Haswell) and this
restatement side by side; tests/test_oracle_golden.py replays the committed vectors.

Every function cites the reference lines it follows (paths relative to /root/reference/make_prg/).
KMeans itself is oracle/kmeans_oracle.c (restating scikit-learn, the reference's third-party dependency).
"""
import ctypes
import gzip
import hashlib
import itertools
import os
import random
import re
import subprocess
from collections import Counter
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

GAP = "-"
NONMATCH = "*"
IUPAC = {"R": "GA", "Y": "TC", "K": "GT", "M": "AC", "S": "GC", "W": "AT", "A": "A", "C": "C", "G": "G", "T": "T"}
ALLOWED = set(IUPAC) | {"N"}
AMBIGUOUS = set("RYKMSW")
MAX_CLUSTERS = 10  # from_msa/cluster_sequences.py:23


class SequenceCurationError(Exception):
    pass


class PartitioningError(Exception):
    pass


Row = Tuple[str, str, str]  # (id, description, sequence)
Alignment = List[Row]

# ----------------------------------------------------------------------------------------------- KMeans binding
_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_kmeans_lib(force=False) -> str:
    src = os.path.join(_HERE, "kmeans_oracle.c")
    out_dir = os.path.join(_HERE, "_build")
    out = os.path.join(out_dir, "libkmeans_oracle.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out, "-lm"])
    return out


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_kmeans_lib())
        _LIB.mprg_oracle_kmeans_fit_predict.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p,
                                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


def kmeans_fit_predict(count_matrix: np.ndarray, k: int, n_init: int = 10, seed: int = 2, want_debug=False):
    """KMeans(n_clusters=k, random_state=2, algorithm='elkan').fit(X).predict(X) — cluster_sequences.py:262-266."""
    X = np.ascontiguousarray(count_matrix, dtype=np.float64)
    D, V = X.shape
    labels = np.zeros(D, np.int32)
    fit_labels = np.zeros(D, np.int32)
    pp = np.zeros(n_init * k, np.int32)
    info = np.zeros(4)
    rc = _lib().mprg_oracle_kmeans_fit_predict(X.ctypes.data, D, V, k, n_init, seed, labels.ctypes.data,
                                               fit_labels.ctypes.data, pp.ctypes.data, info.ctypes.data)
    if rc != 0:
        raise ValueError("kmeans oracle: bad arguments")
    if want_debug:
        return labels, dict(fit_labels=fit_labels, pp=pp.reshape(n_init, k), inertia=info[0], n_iter=int(info[1]),
                            best_restart=int(info[2]), flags=int(info[3]))
    return labels


# ----------------------------------------------------------------------------------------------- ingest (A0)
def parse_fasta(text: str) -> Alignment:
    """Bio.AlignIO.read(handle, 'fasta') as used by utils/io_utils.py:17-31 (id = first word, description = title)."""
    rows, title, chunks = [], None, []
    for line in text.splitlines():
        if line.startswith(">"):
            if title is not None:
                rows.append((title, "".join(chunks)))
            title, chunks = line[1:].rstrip(), []
        elif title is not None:
            chunks.append("".join(line.split()))
    if title is not None:
        rows.append((title, "".join(chunks)))
    if not rows:
        raise ValueError("No records found in handle")
    if len({len(s) for _, s in rows}) != 1:
        raise ValueError("Sequences must all be the same length")
    out = []
    for title, seq in rows:
        words = title.split(None, 1)
        out.append((words[0] if words else "", title, seq))
    return out


def majority_consensus(seqs: List[str]) -> str:
    """utils/seq_utils.py:246-290: per column majority of non-gap non-N residues, ties and empty columns broken
    by random.Random seeded with sha256 of the concatenated upper-cased rows; one RNG draw per column."""
    rng = random.Random()
    rng.seed(hashlib.sha256("".join(seqs).encode()).digest())
    out = []
    ncols = len(seqs[0]) if seqs else 0
    for c in range(ncols):
        counts = Counter(s[c] for s in seqs if s[c] != GAP and s[c] != "N")
        if not counts:
            out.append(rng.choice("ACGT"))
            continue
        top = counts.most_common(1)[0][1]
        out.append(rng.choice([res for res, n in counts.items() if n == top]))
    return "".join(out)


def load_alignment_text(text: str) -> Alignment:
    """utils/io_utils.py:17-49: parse, upper-case, replace every N by the majority-consensus base of its column."""
    rows = [(i, d, s.upper()) for i, d, s in parse_fasta(text)]
    cons = majority_consensus([s for _, _, s in rows])
    return [(i, d, "".join(cons[p] if ch == "N" else ch for p, ch in enumerate(s))) for i, d, s in rows]


def load_alignment_file(path) -> Alignment:
    path = str(path)
    if path.endswith(".gz"):
        with gzip.open(path, "rt") as fh:
            return load_alignment_text(fh.read())
    with open(path) as fh:
        return load_alignment_text(fh.read())


# ----------------------------------------------------------------------------------------------- sequence utils
def ungap(s: str) -> str:
    return s.replace(GAP, "")


def dedupe(seqs):
    seen = set()
    for s in seqs:
        if s not in seen:
            seen.add(s)
            yield s


def expand_sequences(seqs: List[str]) -> List[str]:
    """utils/seq_utils.py:116-153 SequenceExpander.get_expanded_sequences."""
    for s in seqs:
        if not set(s) <= ALLOWED:
            raise SequenceCurationError(f"A slice of a sequence has a disallowed base: {s}")
    out, seen = [], set()
    for s in dedupe(seqs):
        if "N" in s:
            continue
        for combo in itertools.product(*(IUPAC[b] for b in s)):
            e = "".join(combo)
            if e not in seen:
                seen.add(e)
                out.append(e)
    if not out:
        raise SequenceCurationError(f"All sequences in this slice contained N: {seqs}")
    return out


def expanded_from_rows(rows: List[str]) -> List[str]:
    """utils/seq_utils.py:155-158."""
    return expand_sequences([ungap(r) for r in rows])


def consensus_string(rows: List[str]) -> str:
    """utils/seq_utils.py:219-239 get_consensus_from_MSA."""
    out = []
    ncols = len(rows[0]) if rows else 0
    for c in range(ncols):
        col = {r[c] for r in rows} - {"N"}
        if len(col) != 1 or (col & AMBIGUOUS) or col == {GAP}:
            out.append(NONMATCH)
        else:
            out.append(next(iter(col)))
    return "".join(out)


def has_empty_row(rows: List[str], a: int, b: int) -> bool:
    """utils/seq_utils.py:37-42 (closed interval)."""
    return any(all(ch == GAP for ch in r[a:b + 1]) for r in rows)


def drop_all_gap_columns(rows: List[str]) -> List[str]:
    """utils/seq_utils.py:193-216."""
    if not rows:
        return []
    keep = [c for c in range(len(rows[0])) if any(r[c] != GAP for r in rows)]
    return ["".join(r[c] for c in keep) for r in rows]


# ----------------------------------------------------------------------------------------------- intervals (A3-A6)
MATCH, NON = "M", "N"


def partition_intervals(consensus: str, mml: int, rows: List[str]):
    """from_msa/interval_partition.py:81-252.  Returns (match, nonmatch, all) lists of closed (start, stop, type)."""
    match: List[List[int]] = []
    non: List[List[int]] = []

    def add(iv, typ, end=False):
        # interval_partition.py:143-185 _add_interval; returns the interval to continue with, or None
        if typ == MATCH:
            if iv[1] - iv[0] + 1 < mml:
                if non:
                    last = non.pop()
                    last[1] += (iv[1] - iv[0] + 1) + 1
                else:
                    last = [iv[0], iv[1] + 1]
                if end:
                    last[1] -= 1
                    non.append(last)
                return last, NON
            match.append(iv)
            return None
        if match and has_empty_row(rows, iv[0], iv[1]):
            len_match = match[-1][1] - match[-1][0] + 1
            if len_match - 1 < mml:
                match.pop()
                iv[0] -= len_match
                if non:
                    non[-1][1] += iv[1] - iv[0] + 1
                    return None
            else:
                match[-1][1] -= 1
                iv[0] -= 1
        non.append(iv)
        return None

    n = len(consensus)
    if n < mml:
        if n > 0:
            (non if NONMATCH in consensus else match).append([0, n - 1])
    else:
        cur = [0, 0]
        cur_t = NON if consensus[0] == NONMATCH else MATCH
        for i in range(1, n):
            t = NON if consensus[i] == NONMATCH else MATCH
            if t == cur_t:
                cur[1] += 1
            else:
                r = add(cur, cur_t)
                if r is None:
                    cur, cur_t = [i, i], t
                else:
                    cur, cur_t = r
        add(cur, cur_t, end=True)

    # interval_partition.py:187-217 enforce_multisequence_nonmatch_intervals
    if rows:
        for i in reversed(range(len(non))):
            a, b = non[i]
            if len(expanded_from_rows([r[a:b + 1] for r in rows])) < 2:
                match.append([a, b])
                non.pop(i)
    # interval_partition.py:219-252 bijection check (every column of the alignment in exactly one interval)
    ncols = len(rows[0]) if rows else 0
    if ncols:
        cover_m = np.zeros(ncols + 1, np.int64)
        cover_n = np.zeros(ncols + 1, np.int64)
        for cover, ivs in ((cover_m, match), (cover_n, non)):
            for a, b in ivs:
                if a < ncols:
                    cover[max(a, 0)] += 1
                    cover[min(b, ncols - 1) + 1] -= 1
        cm, cn = np.cumsum(cover_m[:ncols]), np.cumsum(cover_n[:ncols])
        bad = (cm > 1) | (cn > 1) | ((cm ^ cn) == 0)
        if bad.any():
            raise PartitioningError(f"Failed interval partitioning at column {int(np.argmax(bad))}")
    m = sorted((a, b, MATCH) for a, b in match)
    nm = sorted((a, b, NON) for a, b in non)
    return m, nm, sorted(m + nm)


# ----------------------------------------------------------------------------------------------- clustering (A9-A12)
def kmer_ids(seqs: List[str], k: int) -> Dict[str, int]:
    """cluster_sequences.py:26-38."""
    for s in seqs:
        if len(s) < k:
            raise ValueError(f"Input sequence {s} has length < kmer size {k}")
    ids: Dict[str, int] = {}
    for s in seqs:
        for p in range(len(s) - k + 1):
            ids.setdefault(s[p:p + k], len(ids))
    return ids


def kmer_count_matrix(seqs: List[str], ids: Dict[str, int]) -> np.ndarray:
    """cluster_sequences.py:41-56."""
    k = len(next(iter(ids)))
    M = np.zeros((len(seqs), len(ids)))
    for j, s in enumerate(seqs):
        for p in range(len(s) - k + 1):
            M[j, ids[s[p:p + k]]] += 1
    return M


def majority_string(seqs: List[str]) -> str:
    """cluster_sequences.py:59-76 (Counter.most_common: ties -> first seen)."""
    if len(seqs) == 1:
        return seqs[0]
    return "".join(Counter(s[c] for s in seqs).most_common(1)[0][0] for c in range(len(seqs[0])))


def one_ref_like(seqs: List[str]) -> bool:
    """cluster_sequences.py:79-104."""
    maj = majority_string(seqs)
    width = len(seqs[0])
    thresh = 1 if width < 5 else int(0.2 * width)
    return all(sum(1 for x, y in zip(s, maj) if x != y) <= thresh for s in seqs)


def needs_more_clusters(clusters: List[List[str]]) -> bool:
    """cluster_sequences.py:107-111."""
    return any(not one_ref_like(c) for c in clusters)


def group_by_label(values: List[list], labels: List[int]) -> List[list]:
    """cluster_sequences.py:114-133 extract_clusters."""
    if len(labels) != len(values):
        raise ValueError("Mismatch between number of sequences/ID lists and number of cluster assignments")
    n = max(labels) + 1
    if set(labels) != set(range(n)):
        raise ValueError("Inconsistent cluster numbering")
    out = [[] for _ in range(n)]
    for lab, v in zip(labels, values):
        out[lab].extend(v)
    return out


def promote_first(clusters: List[List[str]], first_id: str) -> List[List[str]]:
    """cluster_sequences.py:194-208 merge_clusters."""
    rest, chosen = [], []
    for c in clusters:
        if first_id in c:
            chosen = c
        else:
            rest.append(c)
    if not chosen:
        raise ValueError(f"Could not find {first_id} in any cluster")
    chosen = list(chosen)
    chosen.remove(first_id)
    return [[first_id] + chosen] + rest


def merged_alleles(seq_lists, first_seq: str) -> List[str]:
    """cluster_sequences.py:160-191 merge_sequences."""
    others, found = [], False
    for lst in seq_lists:
        for s in lst:
            if s == first_seq:
                found = True
            else:
                others.append(s)
    assert found
    return expand_sequences([first_seq] + others)


@dataclass
class Clustering:
    clustered_ids: List[List[str]]
    sequences: Optional[List[str]] = None
    fits: list = field(default_factory=list)  # [(D, V, k)] KMeans fits executed (for accounting)

    @property
    def no_clustering(self):
        return len(self.clustered_ids) == 1


def cluster_rows(aln: Alignment, k: int, kmeans=kmeans_fit_predict) -> Clustering:
    """cluster_sequences.py:211-296 kmeans_cluster_seqs."""
    long_ids: Dict[str, List[str]] = {}
    long_gapped: Dict[str, List[str]] = {}
    short_ids: Dict[str, List[str]] = {}
    first_id = aln[0][0]
    first_seq = ungap(aln[0][2])
    for rid, _, row in aln:
        s = ungap(row)
        if len(s) >= k:
            long_ids.setdefault(s, []).append(rid)
            long_gapped.setdefault(s, []).append(row)
        else:
            short_ids.setdefault(s, []).append(rid)
    D = len(long_ids)
    fits = []

    def single():
        everything = [x for v in long_ids.values() for x in v] + [x for v in short_ids.values() for x in v]
        return Clustering(promote_first([everything], first_id),
                          merged_alleles([list(long_ids), list(short_ids)], first_seq), fits)

    if D <= 2:
        return single()
    distinct = list(long_ids)
    ids = kmer_ids(distinct, k)
    M = kmer_count_matrix(distinct, ids)
    labels = [0] * D
    groups = group_by_label(list(long_gapped.values()), labels)
    n_clusters = 1
    while needs_more_clusters(groups):
        n_clusters += 1
        if n_clusters > MAX_CLUSTERS:
            break
        if n_clusters == D:
            break
        prev = labels
        labels = [int(x) for x in kmeans(M, n_clusters)]
        fits.append((D, M.shape[1], n_clusters))
        if len(set(labels)) < n_clusters:
            labels = prev
            n_clusters -= 1
            break
        groups = group_by_label(list(long_gapped.values()), labels)
    if n_clusters == 1 or n_clusters == D:
        return single()
    clustered = group_by_label(list(long_ids.values()), labels)
    clustered = promote_first(clustered + [list(v) for v in short_ids.values()], first_id)
    assert sum(len(c) for c in clustered) == len(aln)
    return Clustering(clustered, None, fits)


# ----------------------------------------------------------------------------------------------- recursion (A1,A13-15)
@dataclass
class Node:
    kind: str                 # "leaf" | "interval" | "cluster"
    nesting_level: int
    alignment: Alignment      # stored alignment: all-gap columns removed (recursion_tree.py:45)
    node_id: int
    parent_id: Optional[int]
    children: List["Node"] = field(default_factory=list)


class Builder:
    """prg_builder.py:19-119 state that the recursion and the PRG traversal need."""

    def __init__(self, max_nesting: int, min_match_length: int, kmeans=kmeans_fit_predict):
        self.max_nesting = max_nesting
        self.min_match_length = min_match_length
        self.next_node_id = 0
        self.site_num = 5
        self.prg_index: Dict[Tuple[int, int], int] = {}
        self.kmeans = kmeans
        self.stats = dict(cells_all=0, cells_clustered=0, fits=[])
        self.trace: list = []  # optional per-node records for golden comparison

    def build(self, aln: Alignment, parent: Optional[Node]) -> Node:
        """recursion_tree.py:401-471 NodeFactory.build (+ RecursiveTreeNode.__init__ :33-57)."""
        rows = [r for _, _, r in aln]
        L = self.min_match_length
        m, nm, allv = partition_intervals(consensus_string(rows), L, rows)
        self.stats["cells_all"] += len(rows) * (len(rows[0]) if rows else 0)
        is_leaf = len(allv) == 1 and allv[0][2] == MATCH
        level = 0 if parent is None else parent.nesting_level
        children_alns: List[Alignment] = []
        kind = "leaf"
        if is_leaf:
            pass
        elif len(allv) > 1 or parent is None:
            kind = "interval"
            children_alns = [[(i, d, r[a:b + 1]) for i, d, r in aln] for a, b, _ in allv]
        else:
            self.stats["cells_clustered"] += len(rows) * len(rows[0])
            res = cluster_rows(aln, L, self.kmeans)
            self.stats["fits"].extend(res.fits)
            if self._cluster_further(rows, res, level):
                kind = "cluster"
                level += 1
                children_alns = [[row for row in aln if row[0] in set_ids] for set_ids in res.clustered_ids]
        stored_rows = drop_all_gap_columns(rows)
        node = Node(kind, level, [(i, d, s) for (i, d, _), s in zip(aln, stored_rows)], self.next_node_id,
                    None if parent is None else parent.node_id)
        self.next_node_id += 1
        node.children = [self.build(c, node) for c in children_alns]
        return node

    def _cluster_further(self, rows: List[str], res: Clustering, level: int) -> bool:
        """recursion_tree.py:538-556 + :475-494."""
        if res.no_clustering:
            return False
        if level + 1 >= self.max_nesting:
            return False
        n_ungapped = len(set(ungap(r) for r in rows))
        if n_ungapped <= 2:
            return False
        n_gapped = len(set(rows))
        assert n_ungapped <= n_gapped
        return not n_ungapped < n_gapped

    # ---- PRG emission (recursion_tree.py:194-201, :222-239, :266-300; prg_builder.py:100-119)
    def build_prg(self, root: Node) -> str:
        self.site_num = 5
        self.prg_index = {}
        out: List[str] = []
        self._emit(root, out)
        return "".join(out)

    def _next_site(self) -> int:
        s = self.site_num
        self.site_num += 2
        return s

    def _emit(self, node: Node, out: List[str]):
        if node.kind == "interval":
            for c in node.children:
                self._emit(c, out)
        elif node.kind == "cluster":
            site = self._next_site()
            out.extend(f" {site} ")
            for i, c in enumerate(node.children):
                self._emit(c, out)
                out.extend(f" {site + 1 if i < len(node.children) - 1 else site} ")
        else:
            seqs = expanded_from_rows([r for _, _, r in node.alignment])
            if len(seqs) == 1:
                start = len(out)
                out.extend(seqs[0])
                self.prg_index[(start, len(out))] = node.node_id
            else:
                site = self._next_site()
                out.extend(f" {site} ")
                for i, s in enumerate(seqs):
                    start = len(out)
                    out.extend(s)
                    end = len(out)
                    out.extend(f" {site + 1 if i < len(seqs) - 1 else site} ")
                    self.prg_index[(start, end)] = node.node_id


def tree_dump(root: Node) -> list:
    """Flat preorder dump used for tree-equality (update_DS) checks."""
    out = []

    def rec(n: Node):
        out.append(dict(id=n.node_id, kind=n.kind, level=n.nesting_level, parent=n.parent_id,
                        rows=[[i, s] for i, _, s in n.alignment], children=[c.node_id for c in n.children]))
        for c in n.children:
            rec(c)

    rec(root)
    return out


# ----------------------------------------------------------------------------------------------- encoders (§8f-1)
def encode_prg_ints(prg: str) -> List[int]:
    """utils/prg_encoder.py:44-91."""
    enc = {"A": 1, "C": 2, "G": 3, "T": 4}
    seen: Dict[int, int] = {}
    out: List[int] = []
    for unit in prg.split():
        if all(c.upper() in enc for c in unit):
            out.extend(enc[c.upper()] for c in unit)
        elif unit.isdigit():
            m = int(unit)
            if m % 2 == 1:
                if m not in seen:
                    seen[m] = 1
                    out.append(m)
                else:
                    seen[m] += 1
                    if seen[m] > 2:
                        raise ValueError(f"Prg error: odd site marker {m} found >2 times")
                    out.append(m + 1)
            else:
                out.append(m)
        else:
            raise ValueError(f"Unit {unit} contains invalid characters")
    return out


def encode_prg_bytes(prg: str) -> bytes:
    return np.asarray(encode_prg_ints(prg), dtype="<u4").tobytes()


def gfa_text(prg: str) -> str:
    """utils/gfa.py:16-109."""
    state = dict(text="H\tVN:Z:1.0\tbn:Z:--linear --singlearr\n", gid=0, site=5)

    def split(s: str, site: int) -> List[str]:
        # gfa.py:16-37: re.finditer over the literal " <site> " == non-overlapping leftmost split
        return s.split(f" {site} ")

    def rec(s: str, pre_var_id=None) -> List[int]:
        end_ids: List[int] = []
        while str(state["site"]) in s:
            parts = split(s, state["site"])
            assert len(parts) == 3, f"Invalid prg sequence {s} for site {state['site']}"
            state["text"] += "S\t%d\t%s\tRC:i:0\n" % (state["gid"], parts[0] if parts[0] != "" else "*")
            pre_var_id = state["gid"]
            state["gid"] += 1
            for e in end_ids:
                state["text"] += "L\t%d\t+\t%d\t+\t0M\n" % (e, pre_var_id)
                end_ids = []
            alleles = split(parts[1], state["site"] + 1)
            assert len(alleles) > 1
            state["site"] += 2
            for a in alleles:
                if pre_var_id is not None:
                    state["text"] += "L\t%d\t+\t%d\t+\t0M\n" % (pre_var_id, state["gid"])
                end_ids.extend(rec(a, pre_var_id))
            s = parts[2]
            pre_var_id = None
        state["text"] += "S\t%d\t%s\tRC:i:0\n" % (state["gid"], s if s != "" else "*")
        for e in end_ids:
            state["text"] += "L\t%d\t+\t%d\t+\t0M\n" % (e, state["gid"])
        ret = [state["gid"]]
        state["gid"] += 1
        return ret

    rec(prg)
    return state["text"]


# ----------------------------------------------------------------------------------------------- whole locus
def build_locus(aln: Alignment, max_nesting=5, min_match_length=7, kmeans=kmeans_fit_predict):
    """One locus end to end (subcommands/from_msa.py:108-133 without the files). Returns (prg, builder, root)."""
    b = Builder(max_nesting, min_match_length, kmeans)
    root = b.build(aln, None)
    prg = b.build_prg(root)
    return prg, b, root


def build_locus_from_text(text: str, max_nesting=5, min_match_length=7):
    return build_locus(load_alignment_text(text), max_nesting, min_match_length)

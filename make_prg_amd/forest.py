"""Array-at-a-time host for the batched from_msa build (the throughput path of bench.py and the CLI).

Division of labour: everything that is per ROW or per CELL of an alignment stays on the device (row groups, k-mers,
KMeans, majority/Hamming, children row lists, PRG characters); the Python host only touches per-NODE arrays — a
struct-of-arrays node table that grows one recursion level at a time, vectorised leaf / multi-interval / cluster
classification, per-problem KMeans loop control, and the prefix sums that place every node's text in the PRG string.
The Python host still drives the recursion (north star); it never loops over nodes, rows or characters.

Reference semantics: recursion_tree.py:401-471 (NodeFactory.build), cluster_sequences.py:211-296,
prg_builder.py:100-119 + recursion_tree.py:194-300 (traversals).  Row ids are assumed unique inside an alignment
(the reference partitions cluster children by id); the per-alignment API (engine.BatchEngine) keeps id semantics.
"""
from typing import Dict, List, Optional

import os

import numpy as np

from .backend import MprgError
from .engine import (BIT_GAP, BIT_N, BITS_IUPAC, MAX_CLUSTERS, N_INIT, PF, ROWS_PER_CHUNK, VF, BatchEngine,
                     PartitioningError, SequenceCurationError, expand_sequences)
from .msa import CODE_GAP, decode

KIND_LEAF, KIND_INTERVAL, KIND_CLUSTER = 0, 1, 2
FUSED_VIEWS = os.environ.get("MPRG_FUSED_VIEWS", "1") != "0"     # fused small-view launch shape of mprg_partition
# KMeans rounds: mprg_kmeans_fit with one restart region per problem and one workgroup per fit (restarts + selection in one
# launch; default), or — with MPRG_KMEANS_SLOTS=1 — its persistent form: workgroups that claim fits and keep the per-restart
# arrays in their own scratch slot (0.5 GB of scratch instead of ~3 GB of restart regions per level and worker; measured
# slower on one MI355X: profiles/r02/kmeans_forms.md).  Persistent form: one launch per round as long as a full set of
# resident workgroups (4 per CU) with slots of the round's largest need stays inside SLOT_BUDGET_DOUBLES (2 GiB); fits beyond
# SLOT_SMALL_DOUBLES (2 MiB per slot: config D, Ddeep) get a launch of their own with as many slots as the budget holds.
KMEANS_SLOTS = os.environ.get("MPRG_KMEANS_SLOTS", "0") != "0"
SLOT_SMALL_DOUBLES = 1 << 18
SLOT_BUDGET_DOUBLES = 1 << 28
SLOT_WGS_PER_CU = 4                                              # k_kmeans_fit: 256 threads, 4 waves per SIMD
_ACGT = np.frombuffer(b"ACGT-RYKMSWN????", dtype=np.uint8)


def _excl_cumsum(x: np.ndarray) -> np.ndarray:
    return np.cumsum(x) - x


def _seg_arange(lengths: np.ndarray) -> np.ndarray:
    """0..len-1 inside every segment, concatenated."""
    total = int(lengths.sum())
    return np.arange(total) - np.repeat(_excl_cumsum(lengths), lengths)


def _seg_sum(values: np.ndarray, lengths: np.ndarray) -> np.ndarray:
    """Per-segment sums (segments may be empty)."""
    c = np.concatenate(([0], np.cumsum(values)))
    ends = np.cumsum(lengths)
    return c[ends] - c[ends - lengths]


class Growable:
    """Struct-of-arrays node table that grows level by level."""

    def __init__(self, fields):
        self.fields = fields
        self.chunks = {f: [] for f in fields}
        self.n = 0

    def append(self, **cols):
        k = len(next(iter(cols.values())))
        for f, dt in self.fields.items():
            self.chunks[f].append(np.asarray(cols[f], dtype=dt))
        start = self.n
        self.n += k
        return np.arange(start, self.n)

    def finalize(self):
        return {f: (np.concatenate(c) if c else np.zeros(0, self.fields[f])) for f, c in self.chunks.items()}


class _Offset:
    """A device buffer viewed from a byte offset (only its address is used)."""

    def __init__(self, be, buf, nbytes):
        self.mprg_addr = be.ptr(buf) + int(nbytes)


class ForestEngine(BatchEngine):
    """load() as BatchEngine; run_forest() builds every tree of the batch; assemble_prgs() emits the PRG strings."""

    # ------------------------------------------------------------------------------------------------ device row pool
    def _pool_reserve(self, extra_rows: int):
        need = 4 * (self.pool_used + extra_rows)
        if need > self.pool_cap:
            new_cap = max(2 * self.pool_cap, need, 1 << 16)
            self.d_pool = self.be.grown(self.d_pool, 4 * self.pool_used, new_cap)
            self.pool_cap = new_cap

    def pool_host(self) -> np.ndarray:
        return self.be.download(self.d_pool, np.int32, self.pool_used).astype(np.int64)

    # ------------------------------------------------------------------------------------------------ forest
    def run_forest(self):
        M = len(self._msas)
        meta = np.asarray(self.meta, dtype=np.int64).reshape(M, 6)
        self.meta_arr = meta
        self.failed = np.zeros(M, bool)
        self.errors: Dict[int, Exception] = dict(self.bad)
        for i in self.bad:
            self.failed[i] = True
        # row lists of cluster children live in one device pool; the host keeps only (offset, length) per list
        self.pool_cap, self.pool_used = 0, 0
        self.d_pool = self.be.empty(16)
        self.rl_off = np.zeros(0, np.int64)
        self.rl_len = np.zeros(0, np.int64)
        self.levels: List[dict] = []
        T = Growable(dict(msa=np.int64, parent=np.int64, level=np.int64, rowlist=np.int64, col0=np.int64,
                          ncols=np.int64))
        self.T = T
        self.res_chunks: Dict[str, list] = {k: [] for k in ("kind", "first_child", "n_child", "lvl", "leaf_mode",
                                                            "reps_off", "nseq", "allele_chars", "node_level", "special")}
        ok = np.nonzero(~self.failed)[0]
        cur = dict(msa=ok, parent=np.full(len(ok), -1, np.int64), level=np.zeros(len(ok), np.int64),
                   rowlist=np.full(len(ok), -1, np.int64), col0=np.zeros(len(ok), np.int64), ncols=meta[ok, 5].copy())
        cur["idx"] = T.append(**{k: cur[k] for k in T.fields})
        self.root_of = np.full(M, -1, np.int64)
        self.root_of[ok] = cur["idx"]
        while len(cur["idx"]):
            self.counters["levels"] += 1
            cur = self._forest_level(cur)
        self._finalize_tables()

    # ------------------------------------------------------------------------------------------------ level
    def _view_table_arr(self, cur):
        meta = self.meta_arr
        n = len(cur["idx"])
        tab = np.zeros((n, VF), np.int64)
        m = cur["msa"]
        tab[:, 0:4] = meta[m, 0:4]
        rl = cur["rowlist"]
        has = rl >= 0
        nrows = meta[m, 4].copy()
        rows_off = np.full(n, -1, np.int64)
        if has.any():
            rows_off[has] = self.rl_off[rl[has]]          # offsets into the device row pool
            nrows[has] = self.rl_len[rl[has]]
        tab[:, 4], tab[:, 5], tab[:, 6], tab[:, 7] = rows_off, nrows, cur["col0"], cur["ncols"]
        tab[:, 8] = _excl_cumsum(cur["ncols"])
        tab[:, 9] = _excl_cumsum(nrows)
        return tab

    def _forest_level(self, cur):
        be, L = self.be, self.L
        n = len(cur["idx"])
        # Children that are match intervals straight from their parent's scan (MPRG_IV_PURE) need no kernel: for the same
        # rows every column is one plain base, so the node is a leaf with one allele, its columns as they are
        # (recursion_tree.py:414-420 would find one match interval).  They stay in the level's node range, only the
        # launches skip them.
        pure = cur.get("pure")
        act = np.arange(n) if pure is None else np.nonzero(~pure)[0]
        na = len(act)
        sub_cur = cur if na == n else {k: v[act] for k, v in cur.items()}
        tab_act = self._view_table_arr(sub_cur)
        tab = tab_act if na == n else np.zeros((n, VF), np.int64)
        if na != n:
            tab[act] = tab_act
        n_iv, status, first_type = np.ones(n, np.int64), np.zeros(n, np.int32), np.zeros(n, np.int32)
        has_star, special, iv_off = np.zeros(n, bool), np.zeros(n, bool), np.zeros(n, np.int64)
        iv = np.zeros((0, 3), np.int64)
        if na:
            total_cols = int(tab_act[:, 7].sum())
            cells = float((tab_act[:, 5] * tab_act[:, 7]).sum())
            self.counters["cells_all"] += cells
            d_views, d_rowidx = be.upload(tab_act), self.d_pool
            # small views (the rule below the root): one fused workgroup each, cells in LDS (include/mprg.h); the others
            # go through column masks, gap runs and the scan as separate launches
            Sv, nv = tab_act[:, 5], tab_act[:, 7]
            pitch = (nv + 3) // 4 * 4
            pitch = pitch + np.where((pitch // 4) % 2 == 0, 4, 0)
            fused = (FUSED_VIEWS & (Sv <= 512) & (nv <= 1024) & (Sv * pitch <= 8192) & (nv // max(L - 1, 1) + 4 <= 128)
                     & (Sv > 0) & (nv > 0))
            if fused.any() and na >= 32:
                i_f, i_o = np.nonzero(fused)[0].astype(np.int32), np.nonzero(~fused)[0].astype(np.int32)
                d_if, d_io = be.upload(i_f), be.upload(i_o)
                lists = (be.ptr(d_if), len(i_f), be.ptr(d_io), len(i_o))
            else:
                i_o = np.arange(na, dtype=np.int32)
                lists = (None, 0, None, 0)
            d_mask = be.zeros(4 * total_cols)
            if len(i_o):
                tab_o = tab_act[i_o]
                work, rpc = self._mask_work(tab_o)
                work[:, 0] = i_o[work[:, 0]]                      # work items name views by their row in the uploaded table
                d_work = be.upload(work)
                be.call("mprg_column_masks", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(d_rowidx), be.ptr(d_work),
                        work.shape[0], rpc, be.ptr(d_mask), be.stream, work=float((tab_o[:, 5] * tab_o[:, 7]).sum()))
                wr = self._row_chunk_work(tab_o)
                wr[:, 0] = i_o[wr[:, 0]]
            else:
                wr = np.zeros((0, 2), np.int32)
            d_maxrun, d_stack, d_ivflag = be.zeros(4 * total_cols), be.empty(16 * total_cols), be.zeros(8 * total_cols)
            d_iv, d_niv, d_status = be.empty(12 * total_cols), be.empty(4 * na), be.empty(4 * na)
            d_vout, d_ivp, d_ivc = be.empty(32 * na), be.empty(12 * total_cols), be.zeros(4)
            d_wr = be.upload(wr)
            be.call("mprg_partition", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(d_rowidx), na, be.ptr(d_mask), L,
                    be.ptr(d_wr), len(wr), be.ptr(d_maxrun), be.ptr(d_stack), be.ptr(d_ivflag), be.ptr(d_iv), be.ptr(d_niv),
                    be.ptr(d_status), be.ptr(d_vout), be.ptr(d_ivp), be.ptr(d_ivc), *lists, be.stream, work=cells)
            self.counters["launches"] += 2
            # the column masks and the per-column interval slots stay on the device: the host reads one record per view
            # and the sum(n_iv) interval triples
            vout = be.download(d_vout, np.int32, 8 * na).reshape(na, 8)
            n_iv[act], status[act], first_type[act] = vout[:, 0], vout[:, 1], vout[:, 2]
            has_star[act], special[act], iv_off[act] = (vout[:, 3] & 1) != 0, (vout[:, 3] & 2) != 0, vout[:, 4]
            iv = be.download(d_ivp, np.int32, 3 * int(vout[:, 0].sum())).reshape(-1, 3).astype(np.int64)
        lvl = len(self.levels)
        self.levels.append(dict(idx=cur["idx"]))

        if status.any():          # per-locus policy: the locus is dropped, the batch goes on
            for j in np.nonzero(status)[0]:
                mi = int(cur["msa"][j])
                if not self.failed[mi]:
                    self.failed[mi] = True
                    self.errors[mi] = (SequenceCurationError("All sequences in this slice contained N. Redo sequence curation.")
                                       if status[j] & 2 else PartitioningError("Failed interval partitioning"))
        alive = ~self.failed[cur["msa"]]
        is_leaf = alive & (n_iv == 1) & (first_type == 0)
        is_interval = alive & ~is_leaf & ((n_iv > 1) | (cur["parent"] < 0))
        is_cand = alive & ~is_leaf & ~is_interval

        R = dict(kind=np.full(n, KIND_LEAF, np.int8), first_child=np.full(n, -1, np.int64), n_child=np.zeros(n, np.int64),
                 lvl=np.full(n, lvl, np.int64), leaf_mode=np.zeros(n, np.int8),
                 reps_off=np.full(n, -1, np.int64), nseq=np.ones(n, np.int64), allele_chars=cur["ncols"].copy(),
                 node_level=cur["level"].copy(), special=special)
        R["kind"][is_interval] = KIND_INTERVAL
        nxt = {k: [] for k in ("msa", "parent", "level", "rowlist", "col0", "ncols", "idx", "pure")}

        def add_children(par, rowlist, col0, ncols, level, pure=None):
            cols = dict(msa=cur["msa"][par], parent=cur["idx"][par], level=level, rowlist=rowlist, col0=col0, ncols=ncols)
            idx = self.T.append(**cols)
            for k, v in cols.items():
                nxt[k].append(v)
            nxt["idx"].append(idx)
            nxt["pure"].append(np.zeros(len(idx), bool) if pure is None else pure)
            return idx

        # ---- children of multi-interval nodes: one child per interval, same rows (recursion_tree.py:439-451)
        if is_interval.any():
            pj = np.nonzero(is_interval)[0]
            cnt = n_iv[pj]
            src = np.repeat(iv_off[pj], cnt) + _seg_arange(cnt)
            par = np.repeat(pj, cnt)
            idx = add_children(par, cur["rowlist"][par], cur["col0"][par] + iv[src, 0], iv[src, 1] - iv[src, 0] + 1,
                               cur["level"][par], pure=(iv[src, 2] & 2) != 0)
            R["first_child"][pj] = idx[0] + _excl_cumsum(cnt)
            R["n_child"][pj] = cnt

        # ---- clustering stage (single non-match interval below a non-root node) + row groups of non-trivial leaves
        cands = np.nonzero(is_cand)[0]
        dleaves = np.nonzero(is_leaf & has_star)[0]
        if len(cands) or len(dleaves):
            self._forest_cluster(cur, tab, cands, dleaves, R, add_children, lvl)
        for k, v in R.items():
            self.res_chunks[k].append(v)
        return {k: (np.concatenate(v) if v else np.zeros(0, bool if k == "pure" else np.int64)) for k, v in nxt.items()}

    # ------------------------------------------------------------------------------------------------ clustering
    def _forest_cluster(self, cur, tab, cands, dleaves, R, add_children, lvl):
        be, K = self.be, self.L
        sel = np.concatenate([cands, dleaves])
        ncand, nsel = len(cands), len(cands) + len(dleaves)
        sub = tab[sel].copy()
        S = sub[:, 5]
        usize = S * ((sub[:, 7] + 15) // 16 * 16)          # ungapped rows, row-major, 16-byte pitch
        sub[:, 10] = _excl_cumsum(usize)
        sub[:, 9] = _excl_cumsum(S)
        sub[:, 8] = _excl_cumsum(sub[:, 7])
        tot_rows, tot_u, tot_cols = int(S.sum()), int(usize.sum()), int(sub[:, 7].sum())
        d_sub, d_rowidx = be.upload(sub), self.d_pool
        dd = self._dedupe(d_sub, d_rowidx, nsel, tot_rows, tot_u, work=2.0 * float((S * sub[:, 7]).sum()), sub=sub)
        sm = be.download(dd["summary"], np.int64, 8 * nsel).reshape(nsel, 8)
        n_uu, n_ug, Dq, Tq, sumlen, nshort = (sm[:, i] for i in range(6))
        # every selected view can end as a leaf whose alleles are its distinct rows (recursion_tree.py:272-274): the
        # device keeps the first-appearance lists of this level; the host keeps where they are and how big
        self.levels[lvl]["reps_pos"], self.levels[lvl]["reps_len"] = dd["reps_pos"], dd["reps_len"]
        self.levels[lvl]["reps_rows"] = tot_rows
        R["leaf_mode"][sel] = 1
        R["reps_off"][sel] = sub[:, 9]
        R["nseq"][sel] = n_uu
        R["allele_chars"][sel] = sumlen
        if ncand == 0:
            return
        self.counters["cells_clustered"] += float((S[:ncand] * sub[:ncand, 7]).sum())
        lvl_c = cur["level"][sel]
        # recursion_tree.py:538-556 / :475-494 and cluster_sequences.py:235-246: when the result cannot be used
        leaf_now = (lvl_c + 1 >= self.max_nesting) | (n_uu <= 2) | (n_uu < n_ug) | (Dq <= 2)
        pq = np.nonzero((np.arange(nsel) < ncand) & ~leaf_now)[0]
        if len(pq) == 0:
            return
        d_scratch, d_further = be.empty(12 * tot_cols + 64), be.empty(4 * len(pq))

        def check(act_tab, k, d_labels=None, d_assign=None):
            """cluster_further() of the listed problems on the labels the select step just wrote (k=1: one cluster);
            the same call commits those labels as the problems' accepted assignment."""
            return self._cluster_further(d_sub, d_rowidx, sub, act_tab, k, dd["d_of_row"], d_labels, d_assign, d_scratch,
                                         d_further, d_gcodes=dd["gcodes"])

        # cluster_sequences.py:256: `while cluster_further(...)` is evaluated before any KMeans; a view whose rows are
        # already one-reference-like never uses its k-mer matrix, so the featurisation is only done for the others
        t1 = np.zeros((len(pq), PF), np.int64)
        t1[:, 0], t1[:, 1] = pq, Dq[pq]
        pq = pq[check(t1, 1)]
        P = len(pq)
        if P == 0:
            return
        D, Tp = Dq[pq], Tq[pq]
        so = _excl_cumsum(D)
        cap = np.left_shift(np.int64(1), np.ceil(np.log2(np.maximum(2 * Tp, 16))).astype(np.int64))
        fsz = (Tp + 15) // 16 * 16
        ptab = np.zeros((P, PF), np.int64)
        ptab[:, 0], ptab[:, 1], ptab[:, 2], ptab[:, 3] = pq, D, sub[pq, 9], Tp
        ptab[:, 4], ptab[:, 5], ptab[:, 6], ptab[:, 11] = _excl_cumsum(16 * cap), cap, sub[pq, 9] + pq, _excl_cumsum(fsz)
        d_table, d_flag, d_V = be.empty(int((16 * cap).sum())), be.empty(int(fsz.sum())), be.empty(4 * P)
        d_ptab = be.upload(ptab)
        be.call("mprg_kmer_dictionary", be.ptr(d_sub), be.ptr(d_ptab), P, K, be.ptr(dd["ucodes"]), be.ptr(dd["ulen"]),
                be.ptr(dd["seqrow"]), be.ptr(dd["occ_off"]), be.ptr(d_table), be.ptr(d_flag), be.ptr(d_V), be.stream)
        V = be.download(d_V, np.int32, P).astype(np.int64)
        # the per-restart arrays live in the scratch slots of the persistent workgroups (mprg_kmeans_fit): a problem's
        # workspace holds its common part only (centred matrix, norms, k-means++ tables)
        self._rdoubles = N_INIT * (2 * 10 * V + 2 * D * 10 + 9 * D + 512)           # mprg_kmeans_workspace_doubles, restart part
        wsz = D * V + 2 * V + D + 8 + 4 * D * D + (0 if KMEANS_SLOTS else self._rdoubles)
        ptab[:, 7], ptab[:, 8], ptab[:, 9], ptab[:, 10] = V, _excl_cumsum(D * V), _excl_cumsum(wsz), so
        lo = int(D.sum())
        d_ptab = be.upload(ptab)
        d_x, d_ws = be.zeros(8 * int((D * V).sum())), be.empty(8 * int(wsz.sum()))
        d_labels, d_assign = be.empty(4 * lo), be.zeros(4 * lo)
        be.call("mprg_kmer_counts", be.ptr(d_sub), be.ptr(d_ptab), P, K, be.ptr(dd["ucodes"]), be.ptr(dd["ulen"]),
                be.ptr(dd["seqrow"]), be.ptr(dd["occ_off"]), be.ptr(d_table), be.ptr(d_x), be.stream)
        self._kmeans_prepare(d_ptab, D, V, d_x, d_ws)
        self.counters["launches"] += 3
        d_uni, uoff = self._uniforms_all()
        uoff_arr = np.zeros(MAX_CLUSTERS + 1, np.int64)
        for k_, o_ in uoff.items():
            uoff_arr[k_] = o_

        # cluster_sequences.py:256-274 for all problems of the level, one k per round
        num_clusters = np.ones(P, np.int64)
        active = np.argsort(-(D * V), kind="stable")            # biggest fits first: the grid's tail is its largest problem
        k = 1
        while len(active):                                      # one k per round: the reference's loop as it stands
            k += 1
            num_clusters[active] += 1
            active = active[(num_clusters[active] <= MAX_CLUSTERS) & (num_clusters[active] != D[active])]
            if not len(active):
                break
            # KMeans of the round and, right behind it on the stream, cluster_further on its labels (the device accepts a fit's
            # labels only if they hold k distinct values): one wait for the device per round instead of two
            active, st, info, fur = self._kmeans_round(active, k, D, V, int(uoff_arr[k]), d_ptab, d_uni, d_x, d_ws, d_labels,
                                                       (d_sub, d_rowidx, sub, ptab, dd["d_of_row"], d_assign, d_scratch, dd["gcodes"]))
            nA = len(active)
            if (st & 2).any():
                raise MprgError("KMeans empty-cluster relocation needed NumPy's median-of-medians selection fallback, which "
                                "is not restated on the device; refusing to continue with a possibly different result")
            kb = float((8.0 * D[active] * V[active] * (info[:, 4] + N_INIT)).sum())
            self.counters["fits"] += nA
            self.counters["kmeans_bytes"] += kb
            good = info[:, 3].astype(np.int64) >= k
            num_clusters[active[~good]] -= 1                     # cluster_sequences.py:267-273: revert and stop
            active = active[good & fur]
        # ---- MultiClusterNodes and their children (cluster_sequences.py:276-296, recursion_tree.py:457-469)
        splits = np.nonzero((num_clusters != 1) & (num_clusters != D))[0]
        if not len(splits):
            return
        kfin = np.minimum(num_clusters[splits], MAX_CLUSTERS)
        nchild = kfin + nshort[pq[splits]]
        S_sp = S[pq[splits]]
        self._pool_reserve(int(S_sp.sum()))
        pool_off = self.pool_used + _excl_cumsum(S_sp)
        child_off = _excl_cumsum(nchild)
        sp = np.stack([kfin, pool_off, child_off], axis=1).astype(np.int64)
        d_sizes = be.empty(4 * int(nchild.sum()))
        d_spt, d_spi = be.upload(ptab[splits]), be.upload(sp)
        # rowidx (parents' lists) and pool_out (children's lists) are the same pool, disjoint regions
        be.call("mprg_split_children", be.ptr(d_sub), be.ptr(self.d_pool), be.ptr(d_spt), len(splits), be.ptr(d_spi),
                be.ptr(dd["d_of_row"]), be.ptr(dd["s_of_row"]), be.ptr(d_assign), be.ptr(self.d_pool), be.ptr(d_sizes),
                be.stream)
        self.counters["launches"] += 1
        sizes = be.download(d_sizes, np.int32, int(nchild.sum())).astype(np.int64)
        self.pool_used += int(S_sp.sum())
        base = len(self.rl_len)
        within = np.cumsum(sizes) - sizes - np.repeat(_excl_cumsum(S_sp), nchild)   # offset of a child inside its problem
        self.rl_off = np.concatenate([self.rl_off, np.repeat(pool_off, nchild) + within])
        self.rl_len = np.concatenate([self.rl_len, sizes])
        pj = sel[pq[splits]]                                        # frontier positions of the new cluster nodes
        R["kind"][pj] = KIND_CLUSTER
        R["node_level"][pj] += 1                                    # recursion_tree.py:459
        par = np.repeat(pj, nchild)
        idx = add_children(par, base + np.arange(len(sizes)), cur["col0"][par], cur["ncols"][par], R["node_level"][par])
        R["first_child"][pj] = idx[0] + child_off
        R["n_child"][pj] = nchild

    # ------------------------------------------------------------------------------------------------ KMeans rounds
    def _kmeans_round(self, active, k, D, V, uoff, d_ptab, d_uni, d_x, d_ws, d_labels, cf):
        """One k of the reference's loop (cluster_sequences.py:262-266) for the problems `active`: mprg_kmeans_fit, one
        workgroup per fit (its restarts side by side, then the selection).  With MPRG_KMEANS_SLOTS=1 the persistent form
        of the same entry point — one launch for the fits whose per-restart arrays fit a small slot (~1 000 resident
        workgroups), one with fewer, bigger slots for the rest.
        cluster_further on the round's labels follows on the stream.  Returns (active reordered by launch, status,
        km_info rows, cluster_further answers)."""
        be = self.be
        small = self._rdoubles[active] <= SLOT_SMALL_DOUBLES
        # biggest first (the launch's tail is its largest fit); the persistent form lists its small-slot class first
        order = np.lexsort((-(D[active] * V[active]), ~small)) if KMEANS_SLOTS else np.argsort(-(D[active] * V[active]), kind="stable")
        active, small = active[order], small[order]
        nA, n_small = len(active), int(small.sum())
        ki = np.empty((nA, 5), np.int32)
        ki[:, 0], ki[:, 1], ki[:, 2], ki[:, 3], ki[:, 4] = active, k, 0, uoff, 0
        d_sub, d_rowidx, sub, ptab, d_dor, d_assign, d_scratch, d_gcodes = cf
        # everything the round's launches read goes up in ONE copy, everything the host reads comes back in ONE (a copy is
        # a ~16 us launch of its own on the device and a wait on the host: seven per round were 3 % of the device time)
        act_tab = ptab[active]
        wc, wr, cf_work = self._cluster_further_items(sub, act_tab)
        parts = [act_tab, ki, wc, wr]
        offs, o = [], 0
        for a in parts:
            offs.append(o)
            o += (a.nbytes + 15) & ~15
        packed = np.zeros(max(o, 16), np.uint8)
        for a, at in zip(parts, offs):
            packed[at:at + a.nbytes] = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        d_in = be.upload(packed)
        d_ki = _Offset(be, d_in, offs[1])
        out_info, out_st, out_fur = 0, 64 * nA, 64 * nA + ((4 * nA + 15) & ~15)
        d_out = be.zeros(out_fur + 4 * nA + 16)                       # km_info | km_status (zeroed: the kernels OR into it) | further
        d_info, d_st, d_further = _Offset(be, d_out, out_info), _Offset(be, d_out, out_st), _Offset(be, d_out, out_fur)
        d_next = be.empty(16)
        launch_cf = self._cluster_further_plan(d_sub, d_rowidx, sub, act_tab, k, d_dor, d_labels, d_assign, d_scratch, d_further,
                                               staged=(_Offset(be, d_in, offs[0]), _Offset(be, d_in, offs[2]), len(wc),
                                                       _Offset(be, d_in, offs[3]), len(wr), cf_work), d_gcodes=d_gcodes)
        timed = []                                                      # (entry point, its event slot, rows) when profiling

        def mark(name, rows):
            if be.profile is not None and be.profile.get(name):
                timed.append((name, len(be.profile[name]) - 1, rows))

        if not KMEANS_SLOTS:             # restarts + selection of every fit of the round in one launch, one workgroup per fit
            be.call("mprg_kmeans_fit", be.ptr(d_ptab), be.ptr(d_ki), nA, N_INIT, be.ptr(d_uni), be.ptr(d_x), be.ptr(d_ws), 0, 0, 0, 0,
                    be.ptr(d_labels), be.ptr(d_info), be.ptr(d_st), be.stream)
            mark("mprg_kmeans_fit", slice(0, nA))
            self.counters["launches"] += 1
        for lo, hi in (((0, n_small), (n_small, nA)) if KMEANS_SLOTS else ()):
            n = hi - lo
            if not n:
                continue
            stride = int(self._rdoubles[active[lo:hi]].max())
            n_slots = max(1, min(n, SLOT_WGS_PER_CU * be.n_cus, SLOT_BUDGET_DOUBLES // stride))
            d_slots = be.empty(8 * stride * n_slots)
            off = lambda buf, b: _Offset(be, buf, b * lo)
            be.call("mprg_kmeans_fit", be.ptr(d_ptab), be.ptr(off(d_ki, 20)), n, N_INIT, be.ptr(d_uni), be.ptr(d_x), be.ptr(d_ws),
                    be.ptr(d_slots), stride, n_slots, be.ptr(d_next), be.ptr(d_labels), be.ptr(off(d_info, 64)),
                    be.ptr(off(d_st, 4)), be.stream)
            self.counters["launches"] += 1
            mark("mprg_kmeans_fit", slice(lo, hi))
        launch_cf(d_info)
        raw = be.download(d_out, np.uint8, out_fur + 4 * nA)
        info = raw[:64 * nA].view(np.float64).reshape(nA, 8)
        st = raw[out_st:out_st + 4 * nA].view(np.int32).copy()
        fur = raw[out_fur:out_fur + 4 * nA].view(np.int32).astype(bool)
        for name, ev, rows in timed:         # algorithmic bytes are known only after the fits: 8 D V (Elkan iterations + n_init)
            a0, a1, _ = be.profile[name][ev]
            be.profile[name][ev] = (a0, a1, float((8.0 * D[active[rows]] * V[active[rows]] * (info[rows, 4] + N_INIT)).sum()))
        return active, st, info, fur

    # ------------------------------------------------------------------------------------------------ tables
    def _finalize_tables(self):
        t = self.T.finalize()
        n = self.T.n
        order = np.concatenate([lv["idx"] for lv in self.levels]) if self.levels else np.zeros(0, np.int64)
        for key, chunks in self.res_chunks.items():
            out = np.zeros(n, chunks[0].dtype if chunks else np.int64)
            if chunks:
                out[order] = np.concatenate(chunks)
            t[key] = out
        t["level"] = t.pop("node_level")
        self.tab = t


# ======================================================================================================= PRG assembly
def _digits(v: np.ndarray) -> np.ndarray:
    d = np.ones(v.shape, np.int64)
    for p in (10, 100, 1000, 10000, 100000, 1000000, 10000000, 100000000):
        d += v >= p
    return d


def _write_markers(buf: np.ndarray, pos: np.ndarray, val: np.ndarray):
    """Write ' <val> ' at buf[pos...] for arrays of positions / values."""
    if not len(pos):
        return
    nd = _digits(val)
    buf[pos] = 32
    buf[pos + nd + 1] = 32
    for k in range(int(nd.max())):
        m = nd > k
        buf[pos[m] + 1 + k] = 48 + (val[m] // 10 ** (nd[m] - 1 - k)) % 10


def _special_leaf_alleles(self: "ForestEngine", leaves: np.ndarray) -> Dict[int, List[str]]:
    """Leaves whose columns contain ambiguity codes or N: IUPAC expansion on the host (utils/seq_utils.py:116-153).
    Rare; fetches the leaf's distinct rows from the device lists."""
    t = self.tab
    pool = self.pool_host() if self.pool_used else np.zeros(0, np.int64)
    out: Dict[int, List[str]] = {}
    cache: Dict[int, np.ndarray] = {}
    for lf in leaves.tolist():
        lv = int(t["lvl"][lf])
        if lv not in cache:
            cache[lv] = self.be.download(self.levels[lv]["reps_pos"], np.int32, self.levels[lv]["reps_rows"]).astype(np.int64)
        ro = int(t["reps_off"][lf])
        rp = cache[lv][ro:ro + int(t["nseq"][lf])]
        rl = int(t["rowlist"][lf])
        rows = rp if rl < 0 else pool[self.rl_off[rl] + rp]
        codes = self.codes[int(t["msa"][lf])]
        block = codes[rows, int(t["col0"][lf]):int(t["col0"][lf]) + int(t["ncols"][lf])]
        seqs = [_ACGT[r[r != CODE_GAP]].tobytes().decode() for r in block]
        try:
            out[lf] = expand_sequences(seqs)
        except SequenceCurationError as err:
            mi = int(t["msa"][lf])
            self.failed[mi] = True
            self.errors[mi] = err
            out[lf] = ["A"]
    return out


def assemble_prgs(self: ForestEngine, want_index: bool = False, as_bytes: bool = False):
    """PRG string of every alignment of the batch (None for loci dropped by the curation policy).
    Host (per-node arrays only): preorder ranks and site numbers by prefix sums over the node table, text lengths
    bottom-up, text offsets top-down, cluster-node site markers.  Device: every leaf's alleles and its own markers
    (mprg_leaf_jobs turns leaves into copy jobs from the first-appearance lists, mprg_emit_alleles copies the ungapped
    cells) — the PRG text is ~80 KB per config-C alignment, so this is the byte-heavy part.
    reference: PrgBuilder.build_prg prg_builder.py:100-105; traversals recursion_tree.py:194-201, :222-239, :266-300."""
    be = self.be
    t = self.tab
    n = len(t["msa"])
    M = len(self._msas)
    if n == 0:
        self.prg_index_arrays = (np.zeros(0, np.int64),) * 3
        self.node_id, self.site_count = np.zeros(0, np.int64), np.zeros(M, np.int64)
        return [None] * M
    msa, parent, kind, nch, fch = t["msa"], t["parent"], t["kind"], t["n_child"], t["first_child"]
    meta = self.meta_arr
    leaf_all = kind == KIND_LEAF
    nseq = np.where(leaf_all, t["nseq"], 0)
    achars = np.where(leaf_all, t["allele_chars"], 0)
    # leaves with ambiguity codes / N in their columns: host expansion
    host_leaf: Dict[int, List[str]] = {}
    l1 = np.nonzero(leaf_all & (t["leaf_mode"] == 1))[0]
    if len(l1):
        sp_leaves = l1[t["special"][l1]]                    # the view's masks held N / ambiguity codes (mprg_partition)
        sp_leaves = sp_leaves[~self.failed[msa[sp_leaves]]]
        if len(sp_leaves):
            host_leaf = _special_leaf_alleles(self, sp_leaves)
            for lf, seqs in host_leaf.items():
                nseq[lf] = len(seqs)
                achars[lf] = sum(len(q) for q in seqs)
    valid = ~self.failed[msa]
    nseq[~valid] = 0
    is_leaf = leaf_all & valid
    # ---- preorder rank inside each tree ------------------------------------------------------------------------------
    def add_children_to_parents(val):
        """val[p] += sum of val over p's children, bottom-up (a level's nodes are one contiguous index range and the
        children of a node are contiguous inside it, so the sums are differences of one running sum per level)."""
        for lv in reversed(self.levels[1:]):
            idx = lv["idx"]
            if not len(idx):
                continue
            c = np.concatenate(([0], np.cumsum(val[idx])))
            pp = parent[idx]
            P = pp[idx == fch[pp]]
            lo = fch[P] - idx[0]
            val[P] += c[lo + nch[P]] - c[lo]

    size = np.ones(n, np.int64)
    add_children_to_parents(size)
    pre = np.zeros(n, np.int64)
    for lv in self.levels[1:]:
        idx = lv["idx"]
        if not len(idx):
            continue
        c = np.cumsum(size[idx]) - size[idx]
        pre[idx] = pre[parent[idx]] + 1 + c - c[fch[parent[idx]] - idx[0]]
    # ---- site numbers: openers (cluster nodes, leaves with several alleles) in preorder -------------------------
    opener = valid & ((kind == KIND_CLUSTER) | (is_leaf & (nseq > 1)))
    order = np.lexsort((pre, msa))
    op_sorted = opener[order].astype(np.int64)
    cum = np.cumsum(op_sorted) - op_sorted
    msa_sorted = msa[order]
    first_of_msa = np.ones(n, bool)
    first_of_msa[1:] = msa_sorted[1:] != msa_sorted[:-1]
    base_cum = np.maximum.accumulate(np.where(first_of_msa, cum, 0))
    site = np.zeros(n, np.int64)
    site[order] = 5 + 2 * (cum - base_cum)
    open_len = np.where(opener, _digits(site) + 2, 0)
    mid_len = np.where(opener, _digits(site + 1) + 2, 0)
    n_sites = np.bincount(msa[opener], minlength=M)
    # ---- text lengths bottom-up, starts top-down ---------------------------------------------------------------------
    total = np.where(is_leaf, achars, 0)
    multi = is_leaf & (nseq > 1)
    total[multi] += open_len[multi] * 2 + (nseq[multi] - 1) * mid_len[multi]
    clus = valid & (kind == KIND_CLUSTER)
    total[clus] = open_len[clus] * 2 + (nch[clus] - 1) * mid_len[clus]
    add_children_to_parents(total)                         # nodes of dropped loci carry 0 throughout
    start = np.zeros(n, np.int64)
    roots = self.root_of[~self.failed & (self.root_of >= 0)]
    msa_len = np.zeros(M, np.int64)
    msa_len[msa[roots]] = total[roots]
    msa_base = _excl_cumsum(msa_len)
    start[roots] = msa_base[msa[roots]]
    for lv in self.levels[1:]:
        idx = lv["idx"]
        if not len(idx):
            continue
        p = parent[idx]
        pc = kind[p] == KIND_CLUSTER
        last = idx == fch[p] + nch[p] - 1
        x = total[idx] + np.where(pc, np.where(last, open_len[p], mid_len[p]), 0)
        c = np.cumsum(x) - x
        start[idx] = start[p] + open_len[p] * pc + c - c[fch[p] - idx[0]]
    # ---- device: leaves -> allele copy jobs (+ the leaves' own markers) -> characters ------------------------------------
    total_chars = int(msa_len.sum())
    d_out = be.zeros(total_chars)
    dev_leaf = is_leaf.copy()
    for lf in host_leaf:
        dev_leaf[lf] = False
    dl = np.nonzero(dev_leaf)[0]
    job_off = _excl_cumsum(nseq[dl])
    n_jobs = int(nseq[dl].sum())
    d_jobs = be.empty(32 * max(n_jobs, 1))
    rl = t["rowlist"][dl]
    ltab = np.zeros((len(dl), 10), np.int64)
    ltab[:, 0], ltab[:, 1] = meta[msa[dl], 0], meta[msa[dl], 2]
    ltab[:, 2] = np.where(rl >= 0, self.rl_off[np.maximum(rl, 0)], -1) if len(self.rl_off) else -1
    ltab[:, 3], ltab[:, 4] = t["col0"][dl], t["ncols"][dl]
    ltab[:, 5] = np.where(t["leaf_mode"][dl] == 1, t["reps_off"][dl], -1)
    ltab[:, 6], ltab[:, 7], ltab[:, 8], ltab[:, 9] = nseq[dl], start[dl], np.where(nseq[dl] > 1, site[dl], 0), job_off
    lv_of = np.where(t["leaf_mode"][dl] == 1, t["lvl"][dl], -1)         # -1: needs no level lists
    keep_alive = []
    for lv in np.unique(lv_of):
        m = lv_of == lv
        rp = self.levels[lv]["reps_pos"] if lv >= 0 else None
        rn = self.levels[lv]["reps_len"] if lv >= 0 else None
        d_lt = be.upload(ltab[m])
        keep_alive.append(d_lt)
        be.call("mprg_leaf_jobs", be.ptr(d_lt), int(m.sum()), be.ptr(self.d_pool),
                be.ptr(rp) if rp is not None else None, be.ptr(rn) if rn is not None else None, be.ptr(d_jobs),
                be.ptr(d_out), be.stream)
        self.counters["launches"] += 1
    if n_jobs:
        be.call("mprg_emit_alleles", be.ptr(self.d_arena), be.ptr(d_jobs), n_jobs, be.ptr(d_out), be.stream,
                work=float((nseq[dl] * t["ncols"][dl]).sum() + achars[dl].sum()))
        self.counters["launches"] += 1
    buf = be.download(d_out, np.uint8, total_chars)
    if not buf.flags.writeable:
        buf = buf.copy()
    # ---- host: cluster-node markers, host-expanded leaves --------------------------------------------------------------------
    cn = np.nonzero(clus)[0]
    _write_markers(buf, start[cn], site[cn])
    ch = np.nonzero(valid & (parent >= 0) & (kind[np.maximum(parent, 0)] == KIND_CLUSTER))[0]
    if len(ch):
        p = parent[ch]
        _write_markers(buf, start[ch] + total[ch], np.where(ch == fch[p] + nch[p] - 1, site[p], site[p] + 1))
    host_index = []
    for lf, seqs in host_leaf.items():
        if not valid[lf]:
            continue
        pos = int(start[lf])
        many = len(seqs) > 1
        if many:
            mtxt = f" {site[lf]} ".encode()
            buf[pos:pos + len(mtxt)] = np.frombuffer(mtxt, np.uint8)
            pos += len(mtxt)
        for i, q in enumerate(seqs):
            buf[pos:pos + len(q)] = np.frombuffer(q.encode(), np.uint8)
            host_index.append((lf, pos - int(msa_base[msa[lf]]), pos - int(msa_base[msa[lf]]) + len(q)))
            pos += len(q)
            if many:
                mtxt = f" {site[lf] + 1 if i < len(seqs) - 1 else site[lf]} ".encode()
                buf[pos:pos + len(mtxt)] = np.frombuffer(mtxt, np.uint8)
                pos += len(mtxt)
    out: List[Optional[str]] = [None] * M
    if as_bytes:          # zero-copy views into the batch buffer (ASCII)
        mv = memoryview(buf)
        for i in np.nonzero(~self.failed)[0]:
            out[i] = mv[msa_base[i]:msa_base[i] + msa_len[i]]
    else:
        whole = buf.tobytes()
        for i in np.nonzero(~self.failed)[0]:
            out[i] = whole[msa_base[i]:msa_base[i] + msa_len[i]].decode()
    self.node_id = pre
    self.site_count = n_sites
    if want_index:      # prg_index: every allele of every leaf (recursion_tree.py:276-300)
        jobs = be.download(d_jobs, np.int64, 4 * n_jobs).reshape(-1, 4)
        jl = np.repeat(dl, nseq[dl])
        js = jobs[:, 2] - msa_base[msa[jl]]
        leaf = np.concatenate([jl, np.asarray([h[0] for h in host_index], np.int64)])
        s0 = np.concatenate([js, np.asarray([h[1] for h in host_index], np.int64)])
        s1 = np.concatenate([js + jobs[:, 3], np.asarray([h[2] for h in host_index], np.int64)])
        self.prg_index_arrays = (leaf, s0, s1)
    return out


ForestEngine.assemble_prgs = assemble_prgs


def forest_tree_dump(self: ForestEngine, mi: int, ids: List[str]) -> list:
    """Preorder dump of one tree (same shape as oracle.from_msa_oracle.tree_dump); requires assemble_prgs() first."""
    t = self.tab
    codes = self.codes[mi]
    kinds = {KIND_LEAF: "leaf", KIND_INTERVAL: "interval", KIND_CLUSTER: "cluster"}
    if getattr(self, "_pool_cache_used", -1) != self.pool_used:
        self._pool_cache, self._pool_cache_used = self.pool_host(), self.pool_used
    pool = self._pool_cache
    out = []
    stack = [int(self.root_of[mi])]
    while stack:
        ni = stack.pop()
        rl = int(t["rowlist"][ni])
        rows = np.arange(codes.shape[0]) if rl < 0 else pool[self.rl_off[rl]:self.rl_off[rl] + self.rl_len[rl]]
        c0, w = int(t["col0"][ni]), int(t["ncols"][ni])
        block = codes[rows, c0:c0 + w]
        block = decode(block[:, ~(block == CODE_GAP).all(axis=0)])      # all-gap columns are not stored (recursion_tree.py:45)
        kids = [int(t["first_child"][ni]) + j for j in range(int(t["n_child"][ni]))]
        par = int(t["parent"][ni])
        out.append(dict(id=int(self.node_id[ni]), kind=kinds[int(t["kind"][ni])], level=int(t["level"][ni]),
                        parent=None if par < 0 else int(self.node_id[par]),
                        rows=[[ids[r], b.tobytes().decode()] for r, b in zip(rows, block)],
                        children=[int(self.node_id[c]) for c in kids]))
        stack.extend(reversed(kids))
    return out


def forest_prg_index(self: ForestEngine, mi: int) -> list:
    """[[start, end, node_id], ...] sorted, for one alignment (assemble_prgs(want_index=True) first)."""
    leaf, s, e = self.prg_index_arrays
    m = self.tab["msa"][leaf] == mi
    return sorted([int(a), int(b), int(self.node_id[l])] for l, a, b in zip(leaf[m], s[m], e[m]))


ForestEngine.tree_dump = forest_tree_dump
ForestEngine.prg_index = forest_prg_index

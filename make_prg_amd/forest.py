"""Array-at-a-time host for the batched from_msa build (the throughput path of bench.py and the CLI).

Same device calls and the same decisions as make_prg_amd/engine.py (which keeps the node-object bookkeeping used by
the per-alignment API), but every host step works on whole NumPy arrays over ALL node views of a recursion level:
node table as a struct of arrays, vectorised leaf / multi-interval / cluster classification, segment operations for
the row groups of the clustering stage, and PRG text assembled by prefix sums into one byte buffer.  The Python host
still drives the recursion (north star); it just never loops over nodes.

Reference semantics: recursion_tree.py:401-471 (NodeFactory.build), cluster_sequences.py:211-296,
prg_builder.py:100-119 + recursion_tree.py:194-300 (traversals).  Row ids are assumed unique inside an alignment
(the reference partitions cluster children by id); alignments with duplicate ids are routed to engine.BatchEngine.
"""
from typing import Dict, List, Optional

import numpy as np

from .backend import MprgError
from .engine import (BIT_GAP, BIT_N, BITS_IUPAC, MAX_CLUSTERS, N_INIT, PF, ROWS_PER_CHUNK, VF, BatchEngine,
                     PartitioningError, SequenceCurationError, expand_sequences)
from .msa import CODE_GAP, MSA, decode

KIND_LEAF, KIND_INTERVAL, KIND_CLUSTER = 0, 1, 2
_ACGT = np.frombuffer(b"ACGT-RYKMSWN????", dtype=np.uint8)


def _excl_cumsum(x: np.ndarray) -> np.ndarray:
    return np.cumsum(x) - x


def _seg_ids(lengths: np.ndarray) -> np.ndarray:
    return np.repeat(np.arange(lengths.shape[0]), lengths)


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
            v = cols.get(f)
            self.chunks[f].append(np.full(k, -1, dt) if v is None else np.asarray(v, dtype=dt))
        start = self.n
        self.n += k
        return np.arange(start, self.n)

    def finalize(self):
        return {f: (np.concatenate(c) if c else np.zeros(0, self.fields[f])) for f, c in self.chunks.items()}


class ForestEngine(BatchEngine):
    """load() as BatchEngine; run_forest() builds every tree of the batch with array-at-a-time host code."""

    def run_forest(self):
        be = self.be
        M = len(self._msas)
        meta = np.asarray(self.meta, dtype=np.int64).reshape(M, 6)
        self.meta_arr = meta
        self.failed = np.zeros(M, bool)
        self.errors: Dict[int, Exception] = dict(self.bad)
        for i in self.bad:
            self.failed[i] = True
        self.rl_pool = np.zeros(0, np.int64)      # all row lists of cluster children, concatenated
        self.rl_off = np.zeros(0, np.int64)
        self.rl_len = np.zeros(0, np.int64)
        self.levels: List[dict] = []          # per BFS level: node index range, cons, allgap
        self.reps_pool: List[np.ndarray] = []  # leaf_mode 1: local row positions of distinct rows
        self.reps_ulen_pool: List[np.ndarray] = []
        self.reps_n = 0
        T = Growable(dict(msa=np.int64, parent=np.int64, level=np.int64, rowlist=np.int64, col0=np.int64,
                          ncols=np.int64))
        self.T = T
        # mutable per-node results (filled when the node's level is processed)
        self.kind_c, self.first_child_c, self.n_child_c = [], [], []
        self.lvl_c, self.col_off_c, self.leaf_mode_c, self.reps_off_c, self.reps_cnt_c = [], [], [], [], []
        ok = np.nonzero(~self.failed)[0]
        frontier = T.append(msa=ok, parent=np.full(len(ok), -1), level=np.zeros(len(ok), np.int64),
                            rowlist=np.full(len(ok), -1), col0=np.zeros(len(ok), np.int64), ncols=meta[ok, 5])
        self.root_of = np.full(M, -1, np.int64)
        self.root_of[ok] = frontier
        cur = dict(msa=ok, parent=np.full(len(ok), -1), level=np.zeros(len(ok), np.int64),
                   rowlist=np.full(len(ok), -1), col0=np.zeros(len(ok), np.int64), ncols=meta[ok, 5].copy(), idx=frontier)
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
            rows_off[has] = self.rl_off[rl[has]]          # offsets into the global pool (uploaded whole)
            nrows[has] = self.rl_len[rl[has]]
            rowidx = self.rl_pool.astype(np.int32)
        else:
            rowidx = np.zeros(1, np.int32)
        tab[:, 4], tab[:, 5], tab[:, 6], tab[:, 7] = rows_off, nrows, cur["col0"], cur["ncols"]
        tab[:, 8] = _excl_cumsum(cur["ncols"])
        tab[:, 9] = _excl_cumsum(nrows)
        return tab, rowidx

    def _forest_level(self, cur):
        be, L = self.be, self.L
        n = len(cur["idx"])
        tab, rowidx = self._view_table_arr(cur)
        total_cols = int(tab[:, 7].sum())
        cells = float((tab[:, 5] * tab[:, 7]).sum())
        self.counters["cells_all"] += cells
        d_views, d_rowidx = be.upload(tab), be.upload(rowidx)
        work = self._mask_work(tab)
        d_work, d_mask = be.upload(work), be.zeros(4 * total_cols)
        be.call("mprg_column_masks", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(d_rowidx), be.ptr(d_work),
                work.shape[0], ROWS_PER_CHUNK, be.ptr(d_mask), be.stream, work=cells)
        d_maxrun, d_stack, d_ivflag = be.zeros(4 * total_cols), be.empty(16 * total_cols), be.zeros(8 * total_cols)
        d_iv, d_niv, d_status = be.empty(12 * total_cols), be.empty(4 * n), be.empty(4 * n)
        be.call("mprg_partition", be.ptr(self.d_arena), be.ptr(d_views), be.ptr(d_rowidx), n, be.ptr(d_mask), L,
                be.ptr(d_maxrun), be.ptr(d_stack), be.ptr(d_ivflag), be.ptr(d_iv), be.ptr(d_niv), be.ptr(d_status),
                be.stream, work=cells)
        self.counters["launches"] += 2
        mask = be.download(d_mask, np.uint32, total_cols)
        n_iv = be.download(d_niv, np.int32, n).astype(np.int64)
        status = be.download(d_status, np.int32, n)
        iv = be.download(d_iv, np.int32, 3 * total_cols).reshape(-1, 3).astype(np.int64)

        mm = mask & ~np.uint32(BIT_N)
        single = (mm != 0) & ((mm & (mm - 1)) == 0) & ((mm & BITS_IUPAC) == 0) & (mm != BIT_GAP)
        cons = np.full(total_cols, 255, np.uint8)
        cons[single] = np.log2(mm[single]).astype(np.uint8)
        lvl = len(self.levels)
        self.levels.append(dict(cons=cons, allgap=(mask == BIT_GAP), special=(mask & np.uint32(BITS_IUPAC | BIT_N)) != 0))

        # failures (per-locus policy: the locus is dropped, the batch goes on)
        if status.any():
            for j in np.nonzero(status)[0]:
                mi = int(cur["msa"][j])
                if not self.failed[mi]:
                    self.failed[mi] = True
                    self.errors[mi] = (SequenceCurationError("All sequences in this slice contained N. Redo sequence curation.")
                                       if status[j] & 2 else PartitioningError("Failed interval partitioning"))
        alive = ~self.failed[cur["msa"]]
        col_off = tab[:, 8]
        first_type = iv[np.minimum(col_off, max(total_cols - 1, 0)), 2] if total_cols else np.zeros(n, np.int64)
        is_leaf = alive & (n_iv == 1) & (first_type == 0)
        is_interval = alive & ~is_leaf & ((n_iv > 1) | (cur["parent"] < 0))
        is_cand = alive & ~is_leaf & ~is_interval
        has_star = _seg_sum((cons == 255).astype(np.int64), tab[:, 7]) > 0

        kind = np.full(n, KIND_LEAF, np.int8)
        kind[is_interval] = KIND_INTERVAL
        leaf_mode = np.zeros(n, np.int8)
        leaf_mode[is_leaf & has_star] = 1
        first_child = np.full(n, -1, np.int64)
        n_child = np.zeros(n, np.int64)
        reps_off = np.full(n, -1, np.int64)
        reps_cnt = np.zeros(n, np.int64)
        node_level = cur["level"].copy()

        nxt = dict(msa=[], parent=[], level=[], rowlist=[], col0=[], ncols=[], idx=[])

        # ---- children of multi-interval nodes: one child per interval, same rows (recursion_tree.py:439-451)
        if is_interval.any():
            pj = np.nonzero(is_interval)[0]
            cnt = n_iv[pj]
            src = np.repeat(col_off[pj], cnt) + _seg_arange(cnt)
            par = np.repeat(pj, cnt)
            c_col0 = cur["col0"][par] + iv[src, 0]
            c_ncols = iv[src, 1] - iv[src, 0] + 1
            idx = self.T.append(msa=cur["msa"][par], parent=cur["idx"][par], level=cur["level"][par],
                                rowlist=cur["rowlist"][par], col0=c_col0, ncols=c_ncols)
            first_child[pj] = idx[0] + _excl_cumsum(cnt)
            n_child[pj] = cnt
            for key, val in (("msa", cur["msa"][par]), ("parent", cur["idx"][par]), ("level", cur["level"][par]),
                             ("rowlist", cur["rowlist"][par]), ("col0", c_col0), ("ncols", c_ncols), ("idx", idx)):
                nxt[key].append(val)

        # ---- clustering stage (single non-match interval below a non-root node) + row groups of non-trivial leaves
        cands = np.nonzero(is_cand)[0]
        dleaves = np.nonzero(is_leaf & has_star)[0]
        if len(cands) or len(dleaves):
            self._forest_cluster(cur, tab, d_views, d_rowidx, cands, dleaves, kind, leaf_mode, first_child, n_child,
                                 reps_off, reps_cnt, node_level, nxt)

        self.kind_c.append(kind); self.first_child_c.append(first_child); self.n_child_c.append(n_child)
        self.lvl_c.append(np.full(n, lvl, np.int64)); self.col_off_c.append(col_off.copy())
        self.leaf_mode_c.append(leaf_mode); self.reps_off_c.append(reps_off); self.reps_cnt_c.append(reps_cnt)
        self.levels[lvl]["node_level"] = node_level
        self.levels[lvl]["idx"] = cur["idx"]
        out = {k: (np.concatenate(v) if v else np.zeros(0, np.int64)) for k, v in nxt.items()}
        return out

    # ------------------------------------------------------------------------------------------------ clustering
    def _forest_cluster(self, cur, tab, d_views, d_rowidx, cands, dleaves, kind, leaf_mode, first_child, n_child,
                        reps_off, reps_cnt, node_level, nxt):
        be, K = self.be, self.L
        sel = np.concatenate([cands, dleaves])
        ncand = len(cands)
        sub = tab[sel].copy()
        S = sub[:, 5]
        pad_rows = (S + 15) // 16 * 16
        usize = sub[:, 7] * pad_rows
        sub[:, 10] = _excl_cumsum(usize)
        sub[:, 9] = _excl_cumsum(S)
        sub[:, 8] = _excl_cumsum(sub[:, 7])
        R, tot_u, tot_cols = int(S.sum()), int(usize.sum()), int(sub[:, 7].sum())
        d_sub = be.upload(sub)
        d_ucodes, d_hash = be.empty(tot_u), be.empty(16 * R)
        d_ulen, d_repu, d_repg = be.empty(4 * R), be.empty(4 * R), be.empty(4 * R)
        be.call("mprg_ungap_dedupe", be.ptr(self.d_arena), be.ptr(d_sub), be.ptr(d_rowidx), len(sel), be.ptr(d_ucodes),
                be.ptr(d_hash), be.ptr(d_ulen), be.ptr(d_repu), be.ptr(d_repg), be.stream,
                work=2.0 * float((S * sub[:, 7]).sum()))
        self.counters["launches"] += 1
        ulen = be.download(d_ulen, np.int32, R).astype(np.int64)
        rep_u = be.download(d_repu, np.int32, R).astype(np.int64)
        rep_g = be.download(d_repg, np.int32, R).astype(np.int64)

        seg_start = sub[:, 9]
        row_view = _seg_ids(S)
        local = np.arange(R) - seg_start[row_view]
        is_rep = rep_u == local
        n_uu = _seg_sum(is_rep.astype(np.int64), S)
        n_ug = _seg_sum((rep_g == local).astype(np.int64), S)
        # every selected view records its distinct rows (leaf emission, recursion_tree.py:272-274)
        self.reps_pool.append(local[is_rep])
        self.reps_ulen_pool.append(ulen[is_rep])
        reps_off[sel] = self.reps_n + _excl_cumsum(n_uu)
        reps_cnt[sel] = n_uu
        self.reps_n += int(n_uu.sum())
        leaf_mode[sel] = 1
        if ncand == 0:
            return
        self.counters["cells_clustered"] += float((S[:ncand] * sub[:ncand, 7]).sum())
        iscand_view = np.arange(len(sel)) < ncand
        lvl_c = cur["level"][sel]
        long_rep = is_rep & (ulen >= K)
        Dq = _seg_sum(long_rep.astype(np.int64), S)
        leaf_now = (lvl_c + 1 >= self.max_nesting) | (n_uu <= 2) | (n_uu < n_ug) | (Dq <= 2)
        isprob_view = iscand_view & ~leaf_now
        pq = np.nonzero(isprob_view)[0]
        P = len(pq)
        if P == 0:
            return
        # ---- problems (vectorised over all of them)
        prob_of_view = np.full(len(sel), -1, np.int64)
        prob_of_view[pq] = np.arange(P)
        D = Dq[pq]
        lr_mask = long_rep & isprob_view[row_view]
        lr_rows = np.nonzero(lr_mask)[0]                        # global row index of every long rep, problem-major
        seqrow = local[lr_rows].astype(np.int32)
        so = _excl_cumsum(D)
        occ = ulen[lr_rows] - K + 1
        Tq = _seg_sum(occ, D)
        prob_of_lr = _seg_ids(D)
        occ_cum = np.cumsum(occ) - np.repeat(_excl_cumsum(Tq), D)      # inclusive cumsum inside the problem
        occ_off = np.zeros(int(D.sum()) + P, np.int64)
        occ_off[np.arange(len(lr_rows)) + prob_of_lr + 1] = occ_cum
        cap = np.left_shift(np.int64(1), np.ceil(np.log2(np.maximum(2 * Tq, 16))).astype(np.int64))
        ptab = np.zeros((P, PF), np.int64)
        ptab[:, 0], ptab[:, 1], ptab[:, 2], ptab[:, 3] = pq, D, so, Tq
        ptab[:, 4], ptab[:, 5], ptab[:, 6] = _excl_cumsum(16 * cap), cap, so + np.arange(P)
        fsz = (Tq + 15) // 16 * 16
        ptab[:, 11] = _excl_cumsum(fsz)
        d_seqrow, d_occ = be.upload(seqrow), be.upload(occ_off)
        d_table, d_flag, d_V = be.empty(int((16 * cap).sum())), be.empty(int(fsz.sum())), be.empty(4 * P)
        d_ptab = be.upload(ptab)
        be.call("mprg_kmer_dictionary", be.ptr(d_sub), be.ptr(d_ptab), P, K, be.ptr(d_ucodes), be.ptr(d_ulen),
                be.ptr(d_seqrow), be.ptr(d_occ), be.ptr(d_table), be.ptr(d_flag), be.ptr(d_V), be.stream)
        V = be.download(d_V, np.int32, P).astype(np.int64)
        ptab[:, 7] = V
        ptab[:, 8] = _excl_cumsum(D * V)
        wsz = D * V + 2 * V + D + 8 + N_INIT * (2 * 10 * V + 2 * D * 10 + 9 * D + 512)   # mprg_kmeans_workspace_doubles
        ptab[:, 9] = _excl_cumsum(wsz)
        ptab[:, 10] = so
        lo = int(D.sum())
        d_ptab = be.upload(ptab)
        d_x, d_ws = be.zeros(8 * int((D * V).sum())), be.empty(8 * int(wsz.sum()))
        d_labels, d_info = be.empty(4 * lo), be.empty(64 * P)
        be.call("mprg_kmer_counts", be.ptr(d_sub), be.ptr(d_ptab), P, K, be.ptr(d_ucodes), be.ptr(d_ulen),
                be.ptr(d_seqrow), be.ptr(d_occ), be.ptr(d_table), be.ptr(d_x), be.stream)
        be.call("mprg_kmeans_prepare", be.ptr(d_ptab), P, be.ptr(d_x), be.ptr(d_ws), be.stream)
        self.counters["launches"] += 3

        # rows of problem views: distinct-sequence index of every row, member key (cluster_sequences.py:252-274)
        prow = np.nonzero(isprob_view[row_view])[0]              # global rows of problem views
        long_cum = np.cumsum(long_rep) - long_rep                 # exclusive count of long reps (global)
        d_of_rep = np.where(long_rep, long_cum - long_cum[seg_start[row_view]], -1)
        d_of_row = d_of_rep[seg_start[row_view] + rep_u]          # -1: row's sequence is shorter than k
        order = np.lexsort((local, d_of_row, row_view))
        mkey = np.empty(R, np.int64)
        mkey[order] = np.arange(R) - seg_start[row_view[order]]
        member = (d_of_row >= 0) & isprob_view[row_view]
        mlabel = np.where(member, 0, -1).astype(np.int32)
        d_mkey = be.upload(mkey.astype(np.int32))
        d_scratch, d_further = be.empty(12 * tot_cols + 64), be.empty(4 * P)
        prob_of_row = prob_of_view[row_view]
        label_idx = np.where(member, so[np.maximum(prob_of_row, 0)] + np.maximum(d_of_row, 0), 0)

        def check(act, k):
            d_sp, d_ml = be.upload(ptab[act]), be.upload(mlabel)
            be.call("mprg_cluster_further", be.ptr(self.d_arena), be.ptr(d_sub), be.ptr(d_rowidx), be.ptr(d_sp), len(act),
                    k, be.ptr(d_ml), be.ptr(d_mkey), be.ptr(d_scratch), be.ptr(d_further), be.stream)
            self.counters["launches"] += 1
            return be.download(d_further, np.int32, len(act)).astype(bool)

        num_clusters = np.ones(P, np.int64)
        assign = np.zeros(lo, np.int64)
        prob_of_d = _seg_ids(D)
        active = np.arange(P)[check(np.arange(P), 1)]
        k = 1
        while len(active):
            k += 1
            num_clusters[active] += 1
            active = active[(num_clusters[active] <= MAX_CLUSTERS) & (num_clusters[active] != D[active])]
            if not len(active):
                break
            nA = len(active)
            d_sp, d_st = be.upload(ptab[active]), be.zeros(4 * nA)
            be.call("mprg_kmeans_restarts", be.ptr(d_sp), nA, k, N_INIT, be.ptr(self._uniforms(k)), be.ptr(d_ws),
                    be.ptr(d_st), be.stream)
            be.call("mprg_kmeans_select", be.ptr(d_sp), nA, k, N_INIT, be.ptr(d_x), be.ptr(d_ws), be.ptr(d_labels),
                    be.ptr(d_st), be.ptr(d_info), be.stream)
            self.counters["launches"] += 2
            st = be.download(d_st, np.int32, nA)
            info = be.download(d_info, np.float64, 8 * nA).reshape(nA, 8)
            labels_all = be.download(d_labels, np.int32, lo).astype(np.int64)
            if st.any():
                raise MprgError("KMeans hit an empty cluster (scikit-learn's relocation step is not implemented on "
                                "the device); refusing to continue with a result that may differ from the reference")
            kb = float((8.0 * D[active] * V[active] * (info[:, 4] + N_INIT)).sum())
            self.counters["fits"] += nA
            self.counters["kmeans_bytes"] += kb
            if be.profile is not None and be.profile.get("mprg_kmeans_restarts"):
                a0, a1, _ = be.profile["mprg_kmeans_restarts"][-1]
                be.profile["mprg_kmeans_restarts"][-1] = (a0, a1, kb)
            good = info[:, 3].astype(np.int64) >= k
            num_clusters[active[~good]] -= 1                     # cluster_sequences.py:267-273: revert and stop
            active = active[good]
            if not len(active):
                break
            upd = np.zeros(P, bool)
            upd[active] = True
            dm = upd[prob_of_d]
            assign[dm] = labels_all[dm]
            rm = member & upd[np.maximum(prob_of_row, 0)]
            mlabel[rm] = assign[label_idx[rm]]
            active = active[check(active, k)]

        # ---- results: cluster nodes with children, or leaves (cluster_sequences.py:276-296, recursion_tree.py:457-469)
        splits = np.nonzero((num_clusters != 1) & (num_clusters != D))[0]
        if not len(splits):
            return
        is_split = np.zeros(P, bool)
        is_split[splits] = True
        srow = np.nonzero(is_split[np.maximum(prob_of_row, 0)] & (prob_of_row >= 0))[0]   # rows of splitting views
        kfin = np.zeros(P, np.int64)
        np.maximum.at(kfin, prob_of_d, assign + 1)
        short_rep = is_rep & (ulen < K)
        short_cum = np.cumsum(short_rep) - short_rep
        s_of_rep = np.where(short_rep, short_cum - short_cum[seg_start[row_view]], -1)
        s_of_row = s_of_rep[seg_start[row_view] + rep_u]
        crow = np.where(d_of_row >= 0, assign[label_idx], kfin[np.maximum(prob_of_row, 0)] + s_of_row)
        # the cluster holding the first row goes first (merge_clusters), the others keep label / appearance order
        c0 = np.zeros(P, np.int64)
        first_rows = seg_start[pq]                                 # global row of local position 0 of each problem
        c0[:] = crow[first_rows]
        p_s, c_s, l_s = prob_of_row[srow], crow[srow], local[srow]
        rank = np.where(c_s == c0[p_s], 0, np.where(c_s < c0[p_s], c_s + 1, c_s))
        order = np.lexsort((l_s, rank, p_s))
        p_o, r_o, l_o = p_s[order], rank[order], l_s[order]
        newgrp = np.ones(len(order), bool)
        newgrp[1:] = (p_o[1:] != p_o[:-1]) | (r_o[1:] != r_o[:-1])
        gstart = np.nonzero(newgrp)[0]
        glen = np.diff(np.concatenate((gstart, [len(order)])))
        gprob = p_o[gstart]
        # MSA row indices of the children (views with a row list map local positions through it)
        vj = sel[pq[p_o]]                                          # frontier position of each sorted row's view
        rl = cur["rowlist"][vj]
        if len(self.rl_pool):
            rows_abs = np.where(rl >= 0, self.rl_pool[np.where(rl >= 0, self.rl_off[np.maximum(rl, 0)] + l_o, 0)], l_o)
        else:
            rows_abs = l_o
        base = len(self.rl_len)
        self.rl_off = np.concatenate([self.rl_off, len(self.rl_pool) + gstart])
        self.rl_len = np.concatenate([self.rl_len, glen])
        self.rl_pool = np.concatenate([self.rl_pool, rows_abs])
        par_j = sel[pq[gprob]]                                     # frontier position of the parent of every child
        kind[sel[pq[splits]]] = KIND_CLUSTER
        node_level[sel[pq[splits]]] += 1                           # recursion_tree.py:459
        idx = self.T.append(msa=cur["msa"][par_j], parent=cur["idx"][par_j], level=node_level[par_j],
                            rowlist=base + np.arange(len(gstart)), col0=cur["col0"][par_j], ncols=cur["ncols"][par_j])
        nchild_p = np.bincount(gprob, minlength=P)[splits]
        first_child[sel[pq[splits]]] = idx[0] + _excl_cumsum(nchild_p)
        n_child[sel[pq[splits]]] = nchild_p
        for key, val in (("msa", cur["msa"][par_j]), ("parent", cur["idx"][par_j]), ("level", node_level[par_j]),
                         ("rowlist", base + np.arange(len(gstart))), ("col0", cur["col0"][par_j]),
                         ("ncols", cur["ncols"][par_j]), ("idx", idx)):
            nxt[key].append(val)

    # ------------------------------------------------------------------------------------------------ tables
    def _finalize_tables(self):
        t = self.T.finalize()
        n = self.T.n
        order = np.concatenate([lv["idx"] for lv in self.levels]) if self.levels else np.zeros(0, np.int64)

        def scatter(chunks, dtype):
            out = np.zeros(n, dtype)
            if chunks:
                out[order] = np.concatenate(chunks)
            return out

        t["kind"] = scatter(self.kind_c, np.int8)
        t["first_child"] = scatter(self.first_child_c, np.int64)
        t["n_child"] = scatter(self.n_child_c, np.int64)
        t["lvl"] = scatter(self.lvl_c, np.int64)
        t["col_off"] = scatter(self.col_off_c, np.int64)
        t["leaf_mode"] = scatter(self.leaf_mode_c, np.int8)
        t["reps_off"] = scatter(self.reps_off_c, np.int64)
        t["reps_cnt"] = scatter(self.reps_cnt_c, np.int64)
        lev = np.zeros(n, np.int64)
        if self.levels:
            lev[order] = np.concatenate([lv["node_level"] for lv in self.levels])
        t["level"] = lev
        t["processed"] = np.zeros(n, bool)
        t["processed"][order] = True
        self.tab = t
        self.reps = np.concatenate(self.reps_pool) if self.reps_pool else np.zeros(0, np.int64)
        self.reps_ulen = np.concatenate(self.reps_ulen_pool) if self.reps_ulen_pool else np.zeros(0, np.int64)
        self.special_all = np.concatenate([lv["special"] for lv in self.levels]) if self.levels else np.zeros(0, bool)
        # the concatenated per-level column arrays (consensus codes, all-gap flags)
        offs = _excl_cumsum(np.asarray([len(lv["cons"]) for lv in self.levels], dtype=np.int64)) if self.levels else np.zeros(0, np.int64)
        self.cons_all = np.concatenate([lv["cons"] for lv in self.levels]) if self.levels else np.zeros(0, np.uint8)
        self.allgap_all = np.concatenate([lv["allgap"] for lv in self.levels]) if self.levels else np.zeros(0, bool)
        t["gcol_off"] = offs[t["lvl"]] + t["col_off"] if n else np.zeros(0, np.int64)


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
        # k-th digit from the left
        div = 10 ** (nd[m] - 1 - k)
        buf[pos[m] + 1 + k] = 48 + (val[m] // div) % 10


def assemble_prgs(self: ForestEngine, want_index: bool = False):
    """PRG string of every alignment of the batch (None for loci dropped by the curation policy).
    Host, array-at-a-time: preorder ranks and site numbers by prefix sums over the node table, text offsets by a
    bottom-up length pass and a top-down start pass, site markers scattered with NumPy.  Device: the allele characters
    themselves (mprg_emit_alleles copies the ungapped cells of every allele to its offset) — the PRG text is ~80 KB per
    config-C alignment, so this is the byte-heavy part.
    reference: PrgBuilder.build_prg prg_builder.py:100-105; traversals recursion_tree.py:194-201, :222-239, :266-300."""
    be = self.be
    t = self.tab
    n = len(t["msa"])
    M = len(self._msas)
    if n == 0:
        return [None] * M
    msa, parent, kind, nch, fch = t["msa"], t["parent"], t["kind"], t["n_child"], t["first_child"]
    meta = self.meta_arr
    rl_off = self.rl_off if len(self.rl_off) else np.zeros(1, np.int64)
    rl_pool = self.rl_pool if len(self.rl_pool) else np.zeros(1, np.int64)

    def abs_rows(nodes_idx, local_pos):
        rl = t["rowlist"][nodes_idx]
        return np.where(rl >= 0, rl_pool[np.where(rl >= 0, rl_off[np.maximum(rl, 0)] + local_pos, 0)], local_pos)

    leaf_all = kind == KIND_LEAF
    # ---- alleles: (leaf, source row) pairs ----------------------------------------------------------------------------
    l0 = np.nonzero(leaf_all & (t["leaf_mode"] == 0))[0]           # one allele = any row (all rows equal, no gaps)
    l1 = np.nonzero(leaf_all & (t["leaf_mode"] == 1))[0]           # alleles = the distinct ungapped rows
    cnt1 = t["reps_cnt"][l1]
    p_leaf = np.concatenate([l0, np.repeat(l1, cnt1)])
    rep_idx = np.repeat(t["reps_off"][l1], cnt1) + _seg_arange(cnt1)
    p_local = np.concatenate([np.zeros(len(l0), np.int64), self.reps[rep_idx]])
    p_len = np.concatenate([t["ncols"][l0], self.reps_ulen[rep_idx]])
    p_row = abs_rows(p_leaf, p_local)
    pm = msa[p_leaf]
    p_src = meta[pm, 0] + p_row * meta[pm, 2] + t["col0"][p_leaf]
    p_w = t["ncols"][p_leaf]
    host_chars: Dict[int, np.ndarray] = {}                          # pair index -> ASCII (host-expanded alleles)
    # leaves containing ambiguity codes / N: expansion on the host (utils/seq_utils.py:116-153), rare
    if len(l1):
        sp_cols = np.concatenate(([0], np.cumsum(self.special_all)))
        g = t["gcol_off"][l1]
        leaf_special = (sp_cols[g + t["ncols"][l1]] - sp_cols[g]) > 0
        if leaf_special.any():
            sp_set = set(l1[leaf_special].tolist())
            keep = ~np.isin(p_leaf, l1[leaf_special])
            extra_leaf, extra_len, extra_chars = [], [], []
            for lf in sorted(sp_set):
                sel = np.nonzero(p_leaf == lf)[0]
                seqs = []
                for i in sel:
                    cells = self.host_arena[p_src[i]:p_src[i] + p_w[i]]
                    seqs.append(_ACGT[cells[cells != CODE_GAP]].tobytes().decode())
                try:
                    seqs = expand_sequences(seqs)
                except SequenceCurationError as err:
                    mi = int(msa[lf])
                    self.failed[mi] = True
                    self.errors[mi] = err
                    seqs = ["A"]
                for q in seqs:
                    extra_leaf.append(lf); extra_len.append(len(q)); extra_chars.append(np.frombuffer(q.encode(), np.uint8))
            nkeep = int(keep.sum())
            p_leaf = np.concatenate([p_leaf[keep], np.asarray(extra_leaf, np.int64)])
            p_len = np.concatenate([p_len[keep], np.asarray(extra_len, np.int64)])
            p_src = np.concatenate([p_src[keep], np.full(len(extra_leaf), -1, np.int64)])
            p_w = np.concatenate([p_w[keep], np.zeros(len(extra_leaf), np.int64)])
            for i, ch in enumerate(extra_chars):
                host_chars[nkeep + i] = ch
            o = np.argsort(p_leaf, kind="stable")
            inv = np.empty(len(o), np.int64)
            inv[o] = np.arange(len(o))
            host_chars = {int(inv[i]): ch for i, ch in host_chars.items()}
            p_leaf, p_len, p_src, p_w = p_leaf[o], p_len[o], p_src[o], p_w[o]
        else:
            o = np.argsort(p_leaf, kind="stable")
            p_leaf, p_len, p_src, p_w = p_leaf[o], p_len[o], p_src[o], p_w[o]
    valid = ~self.failed[msa]
    nseq = np.bincount(p_leaf, minlength=n).astype(np.int64)
    nseq[~valid] = 0
    is_leaf = leaf_all & valid
    # ---- preorder rank inside each tree ------------------------------------------------------------------------------
    size = np.ones(n, np.int64)
    for lv in reversed(self.levels):
        idx = lv["idx"]
        p = parent[idx]
        h = p >= 0
        np.add.at(size, p[h], size[idx[h]])
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
    total = np.zeros(n, np.int64)
    np.add.at(total, p_leaf, p_len)
    total[~valid] = 0
    multi = is_leaf & (nseq > 1)
    total[multi] += open_len[multi] * 2 + (nseq[multi] - 1) * mid_len[multi]
    clus = valid & (kind == KIND_CLUSTER)
    total[clus] = open_len[clus] * 2 + (nch[clus] - 1) * mid_len[clus]
    for lv in reversed(self.levels):
        idx = lv["idx"]
        p = parent[idx]
        h = (p >= 0) & valid[idx]
        np.add.at(total, p[h], total[idx[h]])
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
    # ---- allele positions inside their leaves ----------------------------------------------------------------------------
    npair = len(p_leaf)
    newleaf = np.ones(npair, bool)
    newleaf[1:] = p_leaf[1:] != p_leaf[:-1]
    firstpair = np.nonzero(newleaf)[0]
    k_in_leaf = np.arange(npair) - np.repeat(firstpair, np.diff(np.concatenate((firstpair, [npair]))))
    is_multi = nseq[p_leaf] > 1
    lastseq = k_in_leaf == nseq[p_leaf] - 1
    x = p_len + np.where(is_multi, np.where(lastseq, open_len[p_leaf], mid_len[p_leaf]), 0)
    c = np.cumsum(x) - x
    spos = start[p_leaf] + np.where(is_multi, open_len[p_leaf], 0) + c - np.repeat(c[firstpair], np.diff(np.concatenate((firstpair, [npair]))))
    okp = valid[p_leaf]
    # ---- device: copy the allele characters ------------------------------------------------------------------------------
    total_chars = int(msa_len.sum())
    dev = okp & (p_src >= 0)
    jobs = np.stack([p_src[dev], p_w[dev], spos[dev]], axis=1).astype(np.int64)
    d_out = be.zeros(total_chars)
    if len(jobs):
        d_jobs = be.upload(jobs)
        be.call("mprg_emit_alleles", be.ptr(self.d_arena), be.ptr(d_jobs), len(jobs), be.ptr(d_out), be.stream,
                work=float(jobs[:, 1].sum() + p_len[dev].sum()))
        self.counters["launches"] += 1
    buf = be.download(d_out, np.uint8, total_chars).copy()
    for i, chs in host_chars.items():
        if okp[i]:
            buf[spos[i]:spos[i] + len(chs)] = chs
    # ---- host: site markers --------------------------------------------------------------------------------------------------
    cn = np.nonzero(clus)[0]
    _write_markers(buf, start[cn], site[cn])
    ch = np.nonzero(valid & (parent >= 0) & (kind[np.maximum(parent, 0)] == KIND_CLUSTER))[0]
    if len(ch):
        p = parent[ch]
        _write_markers(buf, start[ch] + total[ch], np.where(ch == fch[p] + nch[p] - 1, site[p], site[p] + 1))
    mo = np.nonzero(multi)[0]
    _write_markers(buf, start[mo], site[mo])
    mk = okp & is_multi
    _write_markers(buf, spos[mk] + p_len[mk], np.where(lastseq[mk], site[p_leaf[mk]], site[p_leaf[mk]] + 1))
    out: List[Optional[str]] = [None] * M
    whole = buf.tobytes()
    for i in np.nonzero(~self.failed)[0]:
        out[i] = whole[msa_base[i]:msa_base[i] + msa_len[i]].decode()
    self.node_id = pre
    self.site_count = n_sites
    if want_index:      # prg_index: every allele of every leaf (recursion_tree.py:276-300)
        self.prg_index_arrays = (p_leaf[okp], spos[okp] - msa_base[msa[p_leaf[okp]]],
                                 spos[okp] - msa_base[msa[p_leaf[okp]]] + p_len[okp])
    return out


ForestEngine.assemble_prgs = assemble_prgs


def forest_tree_dump(self: ForestEngine, mi: int, ids: List[str]) -> list:
    """Preorder dump of one tree (same shape as oracle.from_msa_oracle.tree_dump); requires assemble_prgs() first."""
    t = self.tab
    codes = self.codes[mi]
    kinds = {KIND_LEAF: "leaf", KIND_INTERVAL: "interval", KIND_CLUSTER: "cluster"}
    out = []
    stack = [int(self.root_of[mi])]
    while stack:
        ni = stack.pop()
        rl = int(t["rowlist"][ni])
        rows = np.arange(codes.shape[0]) if rl < 0 else self.rl_pool[self.rl_off[rl]:self.rl_off[rl] + self.rl_len[rl]]
        c0, w, g = int(t["col0"][ni]), int(t["ncols"][ni]), int(t["gcol_off"][ni])
        keep = ~self.allgap_all[g:g + w]
        block = decode(codes[rows, c0:c0 + w][:, keep])
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

"""GPU-box helper: what would DENSE child views buy?  Measured, not estimated: at a recursion level whose views
are all small (the fused partition form) the level's views are copied into dense row-major blocks (mprg_compact_columns with a mask
that keeps every column: identity rows, pitch = columns), and the level's mprg_partition and mprg_ungap_dedupe are timed on the
descriptors as the forest builds them (row-index lists into narrow slices of the alignments' rows) against descriptors of the dense
blocks — same kernels, same outputs.  usage: dense_probe.py [alignments] [level]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_batch
from make_prg_amd.backend import make_backend
import make_prg_amd.forest as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7500
target = int(sys.argv[2]) if len(sys.argv) > 2 else 1
F.SPECULATIVE = False
be = make_backend("runtime", 0)
eng = F.ForestEngine(be, 5, 7)
eng.load(make_batch(list(range(n)), 16)[1])
eng.run_forest()
orig = be.call
seen = {"mprg_partition": 0, "mprg_ungap_dedupe": 0}
VF = 12


def dense_views(views_ptr, n_views):
    """(dense buffer, descriptors of the dense blocks) of the n_views views at views_ptr."""
    d_tmp = be.empty(8 * VF * n_views)
    be.lib.mprg_rt_memcpy_async(be.ptr(d_tmp), views_ptr, 8 * VF * n_views, 3, be.stream)
    tab = be.download(d_tmp, np.int64, VF * n_views).reshape(n_views, VF).copy()
    S, C = tab[:, 5], tab[:, 7]
    size = (S * C + 15) // 16 * 16 + 16
    out_off = np.cumsum(size) - size
    tab_c = tab.copy()
    tab_c[:, 8] = np.cumsum(C) - C
    work = eng._row_chunk_work(tab_c, 64)
    d_v, d_w, d_off = be.upload(tab_c), be.upload(work), be.upload(out_off)
    d_mask, d_out, d_kept = be.full(4 * int(C.sum()) + 16, 1), be.empty(int(size.sum()) + 64), be.zeros(4 * n_views)
    orig("mprg_compact_columns", be.ptr(eng.d_arena), be.ptr(d_v), be.ptr(eng.d_pool), be.ptr(d_w), len(work), 64, be.ptr(d_mask), be.ptr(d_out),
         be.ptr(d_off), be.ptr(d_kept), be.stream)
    kept = be.download(d_kept, np.int32, n_views)
    assert (kept == C).all()
    dense = tab.copy()
    dense[:, 0] = be.ptr(d_out) - be.ptr(eng.d_arena) + out_off          # row-major base, relative to the arena pointer the kernels add
    dense[:, 2], dense[:, 4], dense[:, 6] = C, -1, 0                       # pitch = columns, identity rows, first column 0
    return d_out, be.upload(dense), tab


def timed(fn, reps=5):
    fn(); be.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    be.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def call(name, *a, **k):
    if name in seen:
        seen[name] += 1
        if seen[name] == target + 1 + (0 if name == "mprg_partition" else 0):
            if name == "mprg_partition":
                (arena, views, pool, na, mask, L, gw, n_gap, maxrun, stack, ivflag, iv, niv, status, vout, ivp, ivc, fl, n_fused, ol, n_other, stream) = a
                if n_fused:
                    keep, d_dense, tab = dense_views(views, na)
                    # only the views of the fused (small-view) form get dense descriptors; the others keep theirs (they read the transposed copy too)
                    fused = be.download(_Raw(fl), np.int32, n_fused)
                    dense = be.download(d_dense, np.int64, VF * na).reshape(na, VF).copy()
                    sel = np.zeros(na, bool); sel[fused] = True
                    dense[~sel] = tab[~sel]
                    d_dense = be.upload(dense)
                    outs = []
                    for v in (views, be.ptr(d_dense)):
                        bufs = [be.zeros(12 * int(tab[:, 7].sum()) + 64), be.zeros(4 * na), be.zeros(4 * na), be.zeros(32 * na), be.zeros(12 * int(tab[:, 7].sum()) + 64), be.zeros(16)]
                        run = lambda v=v, b=bufs: orig(name, arena, v, pool, na, mask, L, gw, n_gap, maxrun, stack, ivflag, be.ptr(b[0]), be.ptr(b[1]),
                                                       be.ptr(b[2]), be.ptr(b[3]), be.ptr(b[4]), be.ptr(b[5]), fl, n_fused, ol, n_other, stream)
                        ms = timed(run)
                        outs.append((ms, be.download(bufs[1], np.int32, na), be.download(bufs[3], np.int32, 8 * na), be.download(bufs[4], np.int32, 3 * int(tab[:, 7].sum()))))
                    same = all(np.array_equal(x, y) for x, y in zip(outs[0][1:], outs[1][1:]))
                    cells = int((tab[:, 5] * tab[:, 7]).sum())
                    cells = int((tab[sel, 5] * tab[sel, 7]).sum())
                    print(f"level {target}: mprg_partition over {na} views, {n_fused} of them small ({cells / 1e6:.1f} M cells, {int((tab[sel, 4] >= 0).sum())} with row lists): "
                          f"forest descriptors {outs[0][0]:.3f} ms, dense blocks {outs[1][0]:.3f} ms, same outputs: {same}")
            else:
                (arena, views, pool, nsel, K, ddw, n_dd, ucodes, hashes, ulen, rep_u, rep_g, dor, sor, rpos, rlen, seqrow, occ, summary, gcodes, stream) = a
                keep, d_dense, tab = dense_views(views, nsel)
                outs = []
                for v in (views, be.ptr(d_dense)):
                    run = lambda v=v: orig(name, arena, v, pool, nsel, K, ddw, n_dd, ucodes, hashes, ulen, rep_u, rep_g, dor, sor, rpos, rlen, seqrow, occ,
                                           summary, gcodes, stream)
                    ms = timed(run)
                    outs.append((ms, be.download(a[18 - 0] if False else summary_buf(summary, nsel), np.int64, 8 * nsel)))
                cells = int((tab[:, 5] * tab[:, 7]).sum())
                print(f"level {target}: mprg_ungap_dedupe over {nsel} views ({cells / 1e6:.1f} M cells): forest descriptors {outs[0][0]:.3f} ms, "
                      f"dense blocks {outs[1][0]:.3f} ms, same summaries: {np.array_equal(outs[0][1], outs[1][1])}")
    return orig(name, *a, **k)


class _Raw:          # a device address as a buffer for be.download
    def __init__(self, addr): self.mprg_addr = addr


def summary_buf(addr, nsel):
    return _Raw(addr)


be.call = call
eng.run_forest()

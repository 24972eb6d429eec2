"""K6's sample-sample tables (g1 / g4 / gv / ge) by the tiled kernel against the thread-per-element kernel: the same doubles, bit for bit.
Used by the emulated test (CPU) and the GPU test."""
import numpy as np

PF = 12


def tables_of(be, shapes, seed, with_tables):
    """mprg_kmeans_prepare_big over random count matrices of the given (D, V) shapes (with_tables = "lds": mprg_kmeans_prepare's LDS form —
    the matrix staged in LDS, a thread per sample pair); returns per problem the four D x D tables as uint64."""
    rng = np.random.default_rng(seed)
    P = len(shapes)
    ptab = np.zeros((P, PF), np.int64)
    xs, xo, wo = [], 0, 0
    for i, (D, V) in enumerate(shapes):
        X = rng.integers(0, 4, (D, V)).astype(np.float64)
        X[rng.random((D, V)) < 0.6] = 0.0          # k-mer counts are mostly zero
        if D > 2:
            X[D - 1] = X[0]                        # a repeated row: zero distances, equal dot products
        xs.append(X.reshape(-1))
        ptab[i, 1], ptab[i, 7], ptab[i, 8], ptab[i, 9] = D, V, xo, wo
        xo += D * V
        wo += int(be.lib.mprg_kmeans_workspace_doubles(D, V, 10, 1))
    d_p, d_x, d_ws = be.upload(ptab), be.upload(np.concatenate(xs)), be.zeros(8 * wo)
    if with_tables == "lds":
        need = int((8 * (ptab[:, 1] * (ptab[:, 7] | 1) + 2 * ptab[:, 7])).max())
        d_l = be.upload(np.arange(P, dtype=np.int32))
        be.call("mprg_kmeans_prepare", be.ptr(d_p), P, be.ptr(d_x), be.ptr(d_ws), be.ptr(d_l), P, need, None, 0, be.stream)
    else:
        d_xb = be.empty(8 * xo)
        be.call("mprg_kmeans_prepare_big", be.ptr(d_p), be.ptr(d_x), be.ptr(d_ws), None, P, be.ptr(d_xb), with_tables, be.stream)
    be.synchronize()
    ws = be.download(d_ws, np.uint64, wo)
    out = []
    for i, (D, V) in enumerate(shapes):
        base = int(ptab[i, 9]) + D * V + 2 * V + D + 8          # km_ws: Xc mean tmpV xsq scal | g1 g4 gv ge
        out.append([ws[base + t * D * D: base + (t + 1) * D * D].copy() for t in range(4)])
    return out


SHAPES = [(5, 3), (7, 9), (33, 64), (40, 257), (65, 300), (12, 515), (9, 1030), (6, 2049), (34, 2310), (3, 4100), (70, 130), (2, 777)]


def _differences(shapes, old, new, tag=""):
    bad = []
    for (D, V), o, n in zip(shapes, old, new):
        for name, a, b in zip(("g1", "g4", "gv", "ge"), o, n):
            if not np.array_equal(a, b):
                bad.append((D, V, name + tag, int((a != b).sum())))
    return bad


def check_tiled_tables(be, shapes=SHAPES, seed=11):
    old = tables_of(be, shapes, seed, 2)
    return _differences(shapes, old, tables_of(be, shapes, seed, 1))


LDS_SHAPES = [(5, 3), (7, 9), (33, 64), (40, 257), (12, 515), (9, 1030), (6, 2049), (3, 3000), (70, 130), (2, 777), (14, 90), (36, 301)]


def check_lds_tables(be, shapes=LDS_SHAPES, seed=12):
    """K6's LDS form (a thread per sample pair over the matrix in LDS) against the thread-per-element kernel of the global form."""
    old = tables_of(be, shapes, seed, 2)
    return _differences(shapes, old, tables_of(be, shapes, seed, "lds"), " (lds)")

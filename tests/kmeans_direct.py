"""Drive the three KMeans entry points of the C ABI directly on given count matrices (any backend)."""
import numpy as np

PF = 12


def run_kmeans_fits(be, fits, n_init=10, path="global", n_slots=3):
    """path: "global" = mprg_kmeans_restarts + mprg_kmeans_select ("global-nocounts": without the count matrices, i.e. every
    fit reads the centred matrix from its workspace) (one restart region per problem, two launches);
    "one-launch" = mprg_kmeans_fit without scratch slots (a workgroup per fit: restarts, then selection);
    "fit" = mprg_kmeans_fit (persistent workgroups, per-restart arrays in `n_slots` scratch slots, selection fused);
    "wave" = mprg_kmeans_fit_wave by LDS class through fit lists (one wavefront per fit, restart state in LDS); the fits without
    a class take mprg_kmeans_fit with a fit list; "small" = mprg_kmeans_fit_small (128-thread workgroups, trimmed LDS) likewise;
    "lds" = mprg_kmeans_fit_lds (round 6: the restarts' state in LDS, dynamic LDS by class) likewise."""
    groups = {}
    for idx, f in enumerate(fits):
        groups.setdefault(f["k"], []).append(idx)
    results = [None] * len(fits)
    for k, idxs in groups.items():
        P = len(idxs)
        ptab = np.zeros((P, PF), np.int64)
        xs = []
        xo = wo = lo = 0
        for i, idx in enumerate(idxs):
            f = fits[idx]
            D, V = f["shape"]
            X = np.frombuffer(bytes.fromhex(f["counts_i16_hex"]), dtype="<i2").reshape(D, V).astype(np.float64)
            xs.append(X.reshape(-1))
            ptab[i, 1], ptab[i, 7], ptab[i, 8], ptab[i, 9], ptab[i, 10] = D, V, xo, wo, lo
            xo += D * V
            wo += int(be.lib.mprg_kmeans_workspace_doubles(D, V, 10, n_init))
            lo += D
        d_p, d_x, d_ws = be.upload(ptab), be.upload(np.concatenate(xs)), be.empty(8 * wo)
        d_lab = be.empty(4 * lo)
        n_trials = 2 + int(np.log(k))
        d_u = be.upload(be.random_sample(2, n_init * (1 + (k - 1) * n_trials)))
        ki = np.zeros((P, 5), np.int32)
        ki[:, 0], ki[:, 1] = np.arange(P), k
        d_ki = be.upload(ki)
        need = 8 * (ptab[:, 1] * (ptab[:, 7] | 1) + 2 * ptab[:, 7])
        in_lds = (need <= 64 * 1024) & (np.arange(P) % 5 != 4)          # every fifth problem: the global-memory form
        i_l, i_o = np.nonzero(in_lds)[0].astype(np.int32), np.nonzero(~in_lds)[0].astype(np.int32)
        d_il, d_io = be.upload(i_l), be.upload(i_o)
        d_xb = None
        if path in ("wide-stats", "wide-bytes"):          # K6 for big problems: the counts as bytes for the wide fits; wide-stats: no
            d_xb = be.empty(8 * xo)                       # sample-sample tables either (the wide fits compute what their seeding asks for)
            be.call("mprg_kmeans_prepare_big", be.ptr(d_p), be.ptr(d_x), be.ptr(d_ws), None, P, be.ptr(d_xb), 0 if path == "wide-stats" else 1,
                    be.stream)
        else:
            be.call("mprg_kmeans_prepare", be.ptr(d_p), P, be.ptr(d_x), be.ptr(d_ws), be.ptr(d_il), len(i_l),
                    int(need[in_lds].max()) if len(i_l) else 0, be.ptr(d_io), len(i_o), be.stream)
        d_st1, d_info1 = be.zeros(4 * P), be.empty(64 * P)
        if path == "fit":
            W = lambda D, V, r: int(be.lib.mprg_kmeans_workspace_doubles(int(D), int(V), 10, r))
            stride = max(W(ptab[i, 1], ptab[i, 7], n_init) - W(ptab[i, 1], ptab[i, 7], 0) for i in range(P))
            d_slots = be.empty(8 * stride * n_slots)
            be.call("mprg_kmeans_fit", be.ptr(d_p), be.ptr(d_ki), None, P, n_init, be.ptr(d_u), be.ptr(d_x), be.ptr(d_ws),
                    be.ptr(d_slots), stride, n_slots, be.ptr(be.empty(16)), be.ptr(d_lab), be.ptr(d_info1), be.ptr(d_st1), be.stream)
        elif path in ("split", "wide", "wide-stats", "wide-bytes"):          # a workgroup per restart (64 / 1 024 threads), then the selection
            d_l = be.upload(np.arange(P, dtype=np.int32))
            if path == "split":
                be.call("mprg_kmeans_fit_split", be.ptr(d_p), be.ptr(d_ki), be.ptr(d_l), P, n_init, be.ptr(d_u), be.ptr(d_x), be.ptr(d_ws),
                        be.ptr(d_lab), be.ptr(d_info1), be.ptr(d_st1), be.stream)
            else:
                be.call("mprg_kmeans_fit_wide", be.ptr(d_p), be.ptr(d_ki), be.ptr(d_l), P, n_init, be.ptr(d_u), be.ptr(d_x), be.ptr(d_ws),
                        be.ptr(d_lab), be.ptr(d_info1), be.ptr(d_st1), be.ptr(d_xb) if d_xb is not None else None, be.stream)
        elif path == "one-launch":
            be.call("mprg_kmeans_fit", be.ptr(d_p), be.ptr(d_ki), None, P, n_init, be.ptr(d_u), be.ptr(d_x), be.ptr(d_ws), 0, 0, 0, 0,
                    be.ptr(d_lab), be.ptr(d_info1), be.ptr(d_st1), be.stream)
        elif path in ("wave", "small", "lds"):
            classify = ((lambda D, V: be.lib.mprg_kmeans_wave_class(D, V, k)) if path == "wave" else
                        (lambda D, V: be.lib.mprg_kmeans_lds_class(D, V, k, n_init)) if path == "lds" else
                        (lambda D, V: be.lib.mprg_kmeans_small_class(D, V, k, n_init)))
            entry_of = {"wave": "mprg_kmeans_fit_wave", "small": "mprg_kmeans_fit_small", "lds": "mprg_kmeans_fit_lds"}[path]
            cls = np.asarray([classify(int(ptab[i, 1]), int(ptab[i, 7])) for i in range(P)])
            for c in sorted(set(cls.tolist())):
                lst = np.nonzero(cls == c)[0].astype(np.int32)
                d_l = be.upload(lst)
                if c >= 0:
                    be.call(entry_of, be.ptr(d_p), be.ptr(d_ki), be.ptr(d_l), len(lst), c, n_init, be.ptr(d_u), be.ptr(d_x),
                            be.ptr(d_ws), be.ptr(d_lab), be.ptr(d_info1), be.ptr(d_st1), be.stream)
                else:
                    be.call("mprg_kmeans_fit", be.ptr(d_p), be.ptr(d_ki), be.ptr(d_l), len(lst), n_init, be.ptr(d_u), be.ptr(d_x),
                            be.ptr(d_ws), 0, 0, 0, 0, be.ptr(d_lab), be.ptr(d_info1), be.ptr(d_st1), be.stream)
        else:
            be.call("mprg_kmeans_restarts", be.ptr(d_p), be.ptr(d_ki), P, n_init, be.ptr(d_u), 0 if path == "global-nocounts" else be.ptr(d_x),
                    be.ptr(d_ws), be.ptr(d_st1), be.stream)
            be.call("mprg_kmeans_select", be.ptr(d_p), be.ptr(d_ki), P, n_init, be.ptr(d_x), be.ptr(d_ws), be.ptr(d_lab),
                    be.ptr(d_info1), be.stream)
        info = be.download(d_info1, np.float64, 8 * P).reshape(P, 8)
        st = be.download(d_st1, np.int32, P)
        labels = be.download(d_lab, np.int32, lo)
        for i, idx in enumerate(idxs):
            D = fits[idx]["shape"][0]
            o = int(ptab[i, 10])
            results[idx] = dict(labels=labels[o:o + D].tolist(), inertia_hex=float(info[i, 0]).hex(),
                                n_iter=int(info[i, 1]), status=int(st[i]))
    return results

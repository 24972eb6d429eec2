"""Container-only: inputs that drive NumPy's generic arg-introselect (np.argpartition, selection.cpp of the reference's locked
NumPy 1.24; the same code in NumPy 2.x when its x86-simd-sort dispatch is disabled) into its median-of-medians fallback, found
with McIlroy's adversary ("A killer adversary for quicksort", 1999) played against a transcription of the algorithm; checks
the oracle's C restatement against the REAL np.argpartition on them and writes tests/golden/argpartition.json.

    NPY_DISABLE_CPU_FEATURES="AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR AVX2" \
        python -m oracle.tools.gen_argpartition_killer"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def introselect(less, ts, num, kth, stats):
    """Transcription of the arg-introselect (for the adversary only): less(a, b) compares ELEMENT indices."""
    def swap(i, j):
        ts[i], ts[j] = ts[j], ts[i]
    low, high = 0, num - 1
    if kth - low < 3:
        for i in range(kth - low + 1):
            minidx = i
            for k in range(i + 1, high - low + 1):
                if less(ts[low + k], ts[low + minidx]):
                    minidx = k
            swap(low + i, low + minidx)
        return
    if kth == num - 1:
        maxidx = low
        for k in range(low + 1, num):
            if not less(ts[k], ts[maxidx]):
                maxidx = k
        swap(kth, maxidx)
        return
    depth = 2 * (num.bit_length() - 1)
    while low + 1 < high:
        ll, hh = low + 1, high
        if depth > 0 or hh - ll < 5:
            mid = low + (high - low) // 2
            if less(ts[high], ts[mid]): swap(high, mid)
            if less(ts[high], ts[low]): swap(high, low)
            if less(ts[low], ts[mid]): swap(low, mid)
            swap(mid, low + 1)
        else:
            stats["mom"] += 1
            nmed = (hh - ll) // 5
            sub = ll
            for i in range(nmed):
                t = lambda q: ts[sub + q]
                if less(t(1), t(0)): swap(sub + 1, sub)
                if less(t(4), t(3)): swap(sub + 4, sub + 3)
                if less(t(3), t(0)): swap(sub + 3, sub)
                if less(t(4), t(1)): swap(sub + 4, sub + 1)
                if less(t(2), t(1)): swap(sub + 2, sub + 1)
                if less(t(3), t(2)):
                    m = 1 if less(t(3), t(1)) else 3
                else:
                    m = 2
                swap(sub + m, ll + i)
                sub += 5
            if nmed > 2:
                part = ts[ll:ll + nmed]
                introselect(less, part, nmed, nmed // 2, stats)
                ts[ll:ll + nmed] = part
            swap(ll + nmed // 2, low)
            ll, hh = low, high + 1
        depth -= 1
        piv = ts[low]
        while True:
            ll += 1
            while less(ts[ll], piv): ll += 1
            hh -= 1
            while less(piv, ts[hh]): hh -= 1
            if hh < ll:
                break
            swap(hh, ll)
        swap(low, hh)
        if hh >= kth: high = hh - 1
        if hh <= kth: low = ll
    if high == low + 1 and less(ts[high], ts[low]):
        swap(high, low)


def killer(n, kth):
    """Values (floats) for which the selection needs the fallback: McIlroy's adversary."""
    gas = n
    val = [gas] * n
    state = dict(nsolid=0, candidate=0)

    def freeze(x):
        val[x] = state["nsolid"]
        state["nsolid"] += 1

    def less(x, y):
        if val[x] == gas and val[y] == gas:
            freeze(x if x == state["candidate"] else y)
        if val[x] == gas:
            state["candidate"] = x
        elif val[y] == gas:
            state["candidate"] = y
        return val[x] < val[y]

    stats = dict(mom=0)
    introselect(less, list(range(n)), n, kth, stats)
    rest = [i for i in range(n) if val[i] == gas]
    for i in rest:
        freeze(i)
    return np.asarray(val, np.float64), stats["mom"]


def main():
    import oracle.from_msa_oracle as orc
    lib = ctypes.CDLL(orc.build_kmeans_lib())
    lib.mprg_oracle_argpartition.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long]
    cases = []
    for n, kth in ((64, 40), (200, 150), (257, 128), (500, 496), (817, 812), (1000, 333)):
        v, _ = killer(n, kth)
        # with distinct values the transcription replays the same decisions on the frozen values: count the fallbacks it takes
        stats = dict(mom=0)
        introselect(lambda a, b: v[a] < v[b], list(range(n)), n, kth, stats)
        perm = np.arange(n, dtype=np.int64)
        lib.mprg_oracle_argpartition(v.ctypes.data, perm.ctypes.data, n, kth)
        ref = np.argpartition(v, kth)
        assert stats["mom"] > 0, (n, kth, "the adversary did not reach the fallback")
        assert np.array_equal(ref, perm), (n, kth, "the oracle differs from np.argpartition (is the SIMD dispatch disabled?)")
        cases.append(dict(n=n, kth=kth, values=[int(x) for x in v], fallbacks=stats["mom"], argpartition=[int(x) for x in ref]))
        print("n", n, "kth", kth, "median-of-medians rounds", stats["mom"], "oracle == numpy")
    # a duplicate-heavy variant (ties): the same values halved
    for c in list(cases[:3]):
        v = np.asarray(c["values"], np.float64) // 2
        perm = np.arange(len(v), dtype=np.int64)
        lib.mprg_oracle_argpartition(v.ctypes.data, perm.ctypes.data, len(v), c["kth"])
        ref = np.argpartition(v, c["kth"])
        assert np.array_equal(ref, perm)
        cases.append(dict(n=c["n"], kth=c["kth"], values=[int(x) for x in v], fallbacks=None, argpartition=[int(x) for x in ref]))
    with open(os.path.join(ROOT, "tests", "golden", "argpartition.json"), "w") as fh:
        json.dump(dict(numpy=np.__version__, note="np.argpartition with NPY_DISABLE_CPU_FEATURES (generic introselect)", cases=cases), fh)


if __name__ == "__main__":
    main()

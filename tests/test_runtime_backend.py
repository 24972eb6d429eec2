"""HipRuntimeBackend (make_prg_amd/backend.py: the product backend without torch, over the library's mprg_rt_* calls) driven
through the CPU emulation build of the same sources: "device" memory is host memory there, so the allocator, the copies, the
events and the whole recursion forest through this backend are checked in the GPU-less container.  On the GPU,
tests/test_gpu_parity.py runs every case through both product backends."""
import numpy as np
import pytest

from make_prg_amd.backend import HipRuntimeBackend
from tests import parity_common as pc
from tests.emu.backend import build_emu


@pytest.fixture(scope="module")
def rt():
    return HipRuntimeBackend(lib_path=build_emu())


def test_size_classes_and_reuse(rt):
    caps = [rt._capacity(n) for n in (1, 512, 513, 4096, 5000, 1 << 20, (1 << 20) + 1, 3 << 30)]
    assert caps[0] == 512 and caps[1] == 512 and caps[2] == 1024
    for n, c in zip((1, 512, 513, 4096, 5000, 1 << 20, (1 << 20) + 1, 3 << 30), caps):
        assert c >= n and (c - n) <= max(512, n // 8 + 1)
    a = rt.empty(5000)
    addr = a.mprg_addr
    del a
    b = rt.empty(4700)                     # same size class: the block comes back
    assert b.mprg_addr == addr and len(b) == 4700
    z = rt.zeros(1000)
    assert not rt.download(z, np.uint8, 1000).any()


def test_copies_and_events(rt):
    x = np.arange(100000, dtype=np.int64)
    d = rt.upload(x)
    assert np.array_equal(rt.download(d, np.int64, x.size), x)
    g = rt.grown(d, x.nbytes, 2 * x.nbytes)
    assert np.array_equal(rt.download(g, np.int64, x.size), x)
    hb, arr = rt.pinned(1 << 16, "k")
    arr[:1000] = 7
    d2 = rt.upload_from(hb, 1000)
    assert (rt.download(d2, np.uint8, 1000) == 7).all()
    assert rt.pinned(100, "k")[0] is hb                      # kept per key
    hv, harr = rt.host_visible(96 * 8)
    assert harr.size == 96 * 8 and not harr.any() and rt.ptr(hv) == harr.ctypes.data
    rt.async_depth = 2
    outs = []
    for r in range(3):                                       # two buffers used in turn
        got, wait = rt.download_async(d, 8000, group=5)
        wait()
        outs.append(got.ctypes.data)
        assert np.array_equal(got.view(np.int64), x[:1000])
    assert outs[0] == outs[2] != outs[1]
    e0, e1 = rt._event_pair()
    rt._record(e0, None)
    rt._record(e1, None)
    assert e0.elapsed_time(e1) >= 0.0
    rt.trim()
    assert rt._free == {}


def test_forest_through_the_runtime_backend(rt, golden_integration, golden_synthetic):
    assert pc.check_integration(rt, golden_integration) >= 30
    assert pc.check_synthetic(rt, golden_synthetic, configs=("B",), limit=6) == 6


def test_every_thread_is_switched_to_the_backends_device(rt):
    """HIP's current device belongs to the calling thread: a pool thread that drives an engine (bench.py, ranks beyond device 0)
    must be put on the backend's device before its first allocation / launch — once per thread."""
    import threading
    calls = []
    real = rt.lib.mprg_rt_init

    def counting(dev):
        calls.append((threading.get_ident(), dev))
        return real(dev)

    rt.lib.mprg_rt_init = counting
    try:
        def work():
            a = rt.empty(3_000_001)          # a size class nothing else used: a real allocation
            z = rt.zeros(100)
            rt.synchronize()
            assert len(a) == 3_000_001 and not rt.download(z, np.uint8, 100).any()
        t = threading.Thread(target=work)
        t.start(); t.join()
        assert len(calls) == 1 and calls[0][0] != threading.get_ident() and calls[0][1] == rt.device
        rt.empty(64)                         # this thread was switched when the backend was made
        assert len(calls) == 1
    finally:
        rt.lib.mprg_rt_init = real


def test_close_and_regrowth_release_what_the_backend_owns():
    """close() hands free-list blocks, page-locked blocks and streams back; a pinned upload buffer that is outgrown is released
    at once, an outgrown download buffer only after its ring has gone round (a reader may still hold it); one header block
    per backend, not per engine."""
    be = HipRuntimeBackend(lib_path=build_emu())
    freed, hfreed = [], []
    real_free, real_hfree = be.lib.mprg_rt_free, be.lib.mprg_rt_host_free
    be.lib.mprg_rt_free = lambda a: (freed.append(a), real_free(a))[1]
    be.lib.mprg_rt_host_free = lambda a: (hfreed.append(a), real_hfree(a))[1]
    try:
        hb, _ = be.pinned(1 << 20, "arena")
        first = hb.mprg_addr
        hb2, arr2 = be.pinned(4 << 20, "arena")                 # outgrown: the old block goes
        assert hfreed == [first] and arr2.size >= 4 << 20          # (the emulation's malloc may hand the freed address out again)
        assert be.host_visible(768)[0] is be.host_visible(768)[0]
        be.async_depth = 2
        d = be.upload(np.arange(4 << 18, dtype=np.int64))
        small, _ = be.download_async(d, 1 << 10, group=0)
        old = small.ctypes.data
        n_h = len(hfreed)
        big, _ = be.download_async(d, 2 << 20, group=0)          # both ring buffers replaced; the old ones may still be read
        assert len(hfreed) == n_h and small[0] == 0
        be.download_async(d, 2 << 20, group=0)
        assert len(hfreed) == n_h
        be.download_async(d, 2 << 20, group=0)                   # the ring has gone round: now they are un-pinned
        assert len(hfreed) == n_h + 2 and old in hfreed
        a = be.empty(123456)
        addr = a.mprg_addr
        del a
        n_host = len(be._host)
        assert n_host >= 4
        be.close()
        assert addr in freed and len(hfreed) == n_h + 2 + n_host and be._host == [] and be.stream is None
        be.close()                                               # idempotent
    finally:
        be.lib.mprg_rt_free, be.lib.mprg_rt_host_free = real_free, real_hfree

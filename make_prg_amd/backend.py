"""Device backend: loads the HIP shared object through ctypes and owns device memory through PyTorch-ROCm.

The product path has exactly one backend, `HipBackend`: libmprg_hip.so (gfx950 code object) + torch.cuda buffers.
If the library or a GPU is missing this module raises — there is no CPU fallback.  (tests/emu provides a
test-only stand-in with the same interface that runs the kernel source's logic on the CPU; it is injected
explicitly by tests and never imported from here.)
"""
import ctypes
import os
from typing import Optional

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("MPRG_HIP_LIB") or os.path.join(_PKG, "_lib", "libmprg_hip.so")   # MPRG_HIP_LIB: diagnostic builds

c_void_p, c_int, c_int64, c_uint32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32

# name -> (restype, argtypes); mirrors include/mprg.h
SIGNATURES = {
    "mprg_version": (ctypes.c_char_p, []),
    "mprg_last_error": (ctypes.c_char_p, []),
    "mprg_device_cus": (c_int, []),
    "mprg_ingest": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mprg_column_residue_counts": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "mprg_column_masks": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p, c_void_p]),
    "mprg_partition": (c_int, [c_void_p] * 3 + [c_int, c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 9 +
                       [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "mprg_ungap_dedupe": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p, c_int] + [c_void_p] * 14),
    "mprg_kmer_dictionary": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 8),
    "mprg_kmer_counts": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 7),
    "mprg_kmeans_workspace_doubles": (c_int64, [c_int64, c_int64, c_int, c_int]),
    "mprg_kmeans_prepare": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p]),
    "mprg_kmeans_restarts": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mprg_kmeans_fit": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_int64, c_int] + [c_void_p] * 5),
    "mprg_argpartition": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mprg_kmeans_wave_class": (c_int, [c_int64, c_int64, c_int]),
    "mprg_kmeans_fit_wave": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 7),
    "mprg_kmeans_small_class": (c_int, [c_int64, c_int64, c_int, c_int]),
    "mprg_kmeans_fit_small": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 7),
    "mprg_kmeans_select": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 5),
    "mprg_cluster_further": (c_int, [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_int] + [c_void_p] * 6),
    "mprg_split_children": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 7),
    "mprg_leaf_jobs": (c_int, [c_void_p, c_int64] + [c_void_p] * 6),
    "mprg_emit_alleles": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mprg_forest_frontier_count": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_frontier_fill": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_classify": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_children": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_cluster_count": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_cluster_fill": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_problems_count": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_problems_fill": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_sizes_count": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_sizes_fill": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_splits_count": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_splits_fill": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_split_children": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_assemble_special": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_assemble_layout": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_assemble_emit": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_kloop_advance": (c_int, [c_void_p, c_int, c_void_p]),
    "mprg_forest_export_count": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_export_fill": (c_int, [c_void_p, c_void_p]),
    "mprg_random_sample_host": (None, [c_uint32, c_int, c_void_p]),
    "mprg_prg_encode_host": (ctypes.c_longlong, [c_void_p, ctypes.c_longlong, c_void_p]),
    "mprg_fasta_scan_host": (ctypes.c_longlong, [c_void_p, ctypes.c_longlong, c_void_p, c_void_p]),
    "mprg_fasta_fill_host": (ctypes.c_longlong, [c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong, c_void_p]),
    "mprg_gfa_text_host": (ctypes.c_longlong, [c_void_p, ctypes.c_longlong, c_void_p, ctypes.c_longlong]),
    "mprg_ingest_open_host": (c_void_p, [c_void_p, ctypes.c_longlong, c_int]),
    "mprg_ingest_info_host": (None, [c_void_p, c_void_p]),
    "mprg_ingest_fill_host": (None, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "mprg_ingest_text_host": (ctypes.c_longlong, [c_void_p, ctypes.c_longlong, c_void_p]),
    "mprg_ingest_close_host": (None, [c_void_p]),
    "mprg_encode_sizes_host": (None, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mprg_encode_fill_host": (None, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int] + [c_void_p] * 7),
    "mprg_crc32_members_host": (None, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p]),
    "mprg_write_pieces_host": (c_int, [c_int, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int]),
    "mprg_encode_pool_new_host": (c_void_p, []),
    "mprg_encode_pool_reset_host": (None, [c_void_p]),
    "mprg_encode_pool_free_host": (None, [c_void_p]),
    "mprg_encode_pool_info_host": (None, [c_void_p, c_void_p]),
    "mprg_encode_batch_host": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_int] + [c_void_p] * 5),
    "mprg_crc32_host": (c_uint32, [c_uint32, c_void_p, ctypes.c_longlong]),
}


class MprgError(RuntimeError):
    pass


def bind(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


class _Base:
    """Shared call helper: every kernel entry point returns 0 or raises with the library's message."""

    profile = None   # set to a dict to collect per-entry-point device time (HIP events on the launch stream)
    profile_only = None   # optional set of entry points to time (None: all of them)

    def call(self, name, *args, work: float = 0.0, side: Optional[int] = None):
        """Enqueue one C-ABI entry point.  `work` = algorithmic bytes of this launch (roofline accounting).  side: the launch
        goes to side stream `side` (its stream argument must be side_ptr(side)): the timing events are recorded there."""
        ev = self._event_pair() if self.profile is not None and (self.profile_only is None or name in self.profile_only) else None
        if ev:
            self._record(ev[0], side)
        rc = getattr(self.lib, name)(*args)
        if ev:
            self._record(ev[1], side)
            self.profile.setdefault(name, []).append((ev[0], ev[1], float(work)))
        if rc != 0:
            raise MprgError(f"{name} failed ({rc}): {self.lib.mprg_last_error().decode()}")

    def _event_pair(self):
        return None

    def _record(self, event, side):
        event.record()

    # side streams: independent launches of one step side by side (fork: they wait for what the main stream holds so far;
    # join: the main stream waits for them).  The base class has none: everything stays on the one stream.
    n_side_streams = 0

    def side_ptr(self, i: int):
        return self.stream

    def fork(self, n: int):
        pass

    def join(self, n: int):
        pass

    def profile_summary(self):
        """{entry point: dict(calls, ms, bytes)} from the recorded events (synchronises)."""
        self.synchronize()
        out = {}
        for name, evs in (self.profile or {}).items():
            ms = sum(a.elapsed_time(b) for a, b, _ in evs)
            out[name] = dict(calls=len(evs), ms=ms, bytes=sum(w for _, _, w in evs))
        return out

    def random_sample(self, seed: int, n: int) -> np.ndarray:
        out = np.empty(n, np.float64)
        self.lib.mprg_random_sample_host(seed, n, out.ctypes.data)
        return out


class HipBackend(_Base):
    name = "hip"

    def __init__(self, device: Optional[int] = None, lib_path: str = HIP_LIB_PATH, own_stream: bool = False):
        import torch
        if not os.path.exists(lib_path):
            raise MprgError(f"{lib_path} not found: build it with `python __graft_entry__.py build` (hipcc, gfx950)")
        if not torch.cuda.is_available():
            raise MprgError("no ROCm device visible: make_prg_amd has no CPU fallback")
        self.torch = torch
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.lib = bind(ctypes.CDLL(lib_path))
        # own_stream: a private HIP stream, so that several host threads (each with its own backend + engine) overlap
        # their kernels, copies and host work; use `with backend.on_stream():` around the work of that thread
        self.stream_obj = torch.cuda.Stream(self.device) if own_stream else torch.cuda.current_stream(self.device)
        self.stream = self.stream_obj.cuda_stream
        self.n_cus = self.lib.mprg_device_cus()

    def on_stream(self):
        return self.torch.cuda.stream(self.stream_obj)

    # buffers are flat uint8 tensors; sizes in bytes
    def empty(self, nbytes: int):
        return self.torch.empty(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)

    def zeros(self, nbytes: int):
        return self.torch.zeros(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes == 0:
            return self.empty(16)
        t = self.torch.from_numpy(arr.view(np.uint8).reshape(-1))
        return t.to(self.device, non_blocking=False)

    def download(self, buf, dtype, count: int) -> np.ndarray:
        nbytes = int(count) * np.dtype(dtype).itemsize
        if nbytes == 0:
            return np.empty(0, dtype)
        return buf[:nbytes].cpu().numpy().view(dtype)

    def pinned(self, nbytes: int, key):
        """(buffer, uint8 array over it): page-locked host memory for uploads, kept and reused per `key` (grown when needed:
        page-locking is slow, ~1 s per 6 GB)."""
        if not hasattr(self, "_pinned_up"):
            self._pinned_up = {}
        t = self._pinned_up.get(key)
        if t is None or t.numel() < nbytes:
            t = self._pinned_up[key] = self.torch.empty(max(int(nbytes) + (int(nbytes) >> 3), 1 << 20), dtype=self.torch.uint8, pin_memory=True)
        return t, t.numpy()

    def upload_from(self, pinned_buf, nbytes: int):
        """Device copy of the first nbytes of a pinned() buffer, enqueued on the compute stream (no staging copy)."""
        if nbytes == 0:
            return self.empty(16)
        return pinned_buf[:int(nbytes)].to(self.device, non_blocking=True)

    def host_visible(self, nbytes: int):
        """(buffer whose ptr() kernels may write, uint8 array over the same memory): page-locked host memory — hipHostMalloc
        memory is mapped into the device's address space — for results a few words long: the device stores them, the host waits
        for the stream and reads; no copy is enqueued."""
        t = self.torch.zeros(int(nbytes), dtype=self.torch.uint8, pin_memory=True)
        return t, t.numpy()

    def download_async(self, buf, nbytes: int, group: int = 0):
        """Start copying buf[:nbytes] into a PINNED host buffer on the backend's copy stream, behind everything enqueued on the
        compute stream so far; returns (uint8 array over the pinned buffer, wait()).  The array's contents are valid after
        wait(); successive calls of one `group` cycle through `async_depth` (2) pinned buffers, so the copy of one batch overlaps
        the kernels of the next and a result stays valid until the async_depth-th following call of that group.
        utils/io_utils.py:105-110 writes these bytes to files."""
        torch = self.torch
        nbytes = int(nbytes)
        if not hasattr(self, "_pinned"):
            self._pinned, self._copy_stream, self._parity = {}, torch.cuda.Stream(self.device), {}
        depth = getattr(self, "async_depth", 2)          # buffers a group cycles through (a pipeline with three stages sets 3)
        par = self._parity.get(group, 0) % depth
        self._parity[group] = par + 1
        host = self._pinned.get((group, par))
        if host is None or host.numel() < nbytes:
            # all buffers at once: page-locking GBs takes longer than a whole batch, better paid during warm-up
            for q in range(depth):
                if self._pinned.get((group, q)) is None or self._pinned[(group, q)].numel() < nbytes:
                    self._pinned[(group, q)] = torch.empty(max(nbytes + (nbytes >> 3), 1 << 20), dtype=torch.uint8, pin_memory=True)
            host = self._pinned[(group, par)]
        self._copy_stream.wait_stream(self.stream_obj)
        with torch.cuda.stream(self._copy_stream):
            if nbytes:
                host[:nbytes].copy_(buf[:nbytes], non_blocking=True)
            buf.record_stream(self._copy_stream)          # the caching allocator must not hand buf out before the copy ran
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        return host[:nbytes].numpy(), done.synchronize

    def ptr(self, buf) -> int:
        return buf.mprg_addr if hasattr(buf, "mprg_addr") else buf.data_ptr()

    def grown(self, buf, used_bytes: int, new_bytes: int):
        """A larger buffer holding the first used_bytes of buf (device-to-device copy)."""
        new = self.torch.empty(int(new_bytes), dtype=self.torch.uint8, device=self.device)
        if used_bytes:
            new[:used_bytes].copy_(buf[:used_bytes])
        return new

    def synchronize(self):
        self.stream_obj.synchronize()

    def _event_pair(self):
        # torch.cuda.Event wraps hipEvent_t; kernels are enqueued on the backend's stream, the one these record on
        return (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))

    def _record(self, event, side):
        event.record(self.stream_obj if side is None else self._side[side])

    n_side_streams = 3

    def _sides(self, n):
        if not hasattr(self, "_side"):
            self._side = []
        while len(self._side) < n:
            self._side.append(self.torch.cuda.Stream(self.device))
        return self._side[:n]

    def side_ptr(self, i: int):
        return self._sides(i + 1)[i].cuda_stream

    def fork(self, n: int):
        ev = self.torch.cuda.Event()
        ev.record(self.stream_obj)
        for s_ in self._sides(n):
            s_.wait_event(ev)

    def join(self, n: int):
        for s_ in self._sides(n):
            ev = self.torch.cuda.Event()
            ev.record(s_)
            self.stream_obj.wait_event(ev)

"""Device backend: loads the HIP shared object through ctypes and owns device memory through PyTorch-ROCm.

The product path has exactly one backend, `HipBackend`: libmprg_hip.so (gfx950 code object) + torch.cuda buffers.
If the library or a GPU is missing this module raises — there is no CPU fallback.  (tests/emu provides a
test-only stand-in with the same interface that runs the kernel source's logic on the CPU; it is injected
explicitly by tests and never imported from here.)
"""
import ctypes
import os
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run one after the other.
# A worker here drives two engines, each with a compute, a side and a copy stream: with eight queues a 3 750-alignment pass took
# 49.7 ms against 50.9 (first pass, profiles/r05/shards/), four engines 50.5 against 67.7.  Read by the runtime when it starts: set
# before the library (or torch) is loaded; an explicit setting of the caller's wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
    "mprg_compact_columns": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mprg_partition": (c_int, [c_void_p] * 3 + [c_int, c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 9 +
                       [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "mprg_ungap_dedupe": (c_int, [c_void_p] * 3 + [c_int, c_int, c_void_p, c_int] + [c_void_p] * 14),
    "mprg_kmer_dictionary": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 8),
    "mprg_kmer_dictionary_parts": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 7 + [c_int, c_void_p, c_void_p]),
    "mprg_kmer_counts": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 7),
    "mprg_kmer_counts_parts": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 6 + [c_int, c_void_p]),
    "mprg_kmeans_workspace_doubles": (c_int64, [c_int64, c_int64, c_int, c_int]),
    "mprg_kmeans_prepare": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p]),
    "mprg_kmeans_restarts": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mprg_kmeans_fit": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_int64, c_int] + [c_void_p] * 5),
    "mprg_argpartition": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mprg_kmeans_fit_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 7),
    "mprg_kmeans_prepare_big": (c_int, [c_void_p] * 4 + [c_int, c_void_p, c_int, c_void_p]),
    "mprg_kmeans_fit_wide": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 8),
    "mprg_kmeans_wave_class": (c_int, [c_int64, c_int64, c_int]),
    "mprg_kmeans_fit_wave": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 7),
    "mprg_kmeans_small_class": (c_int, [c_int64, c_int64, c_int, c_int]),
    "mprg_kmeans_fit_small": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 7),
    "mprg_kmeans_lds_class": (c_int, [c_int64, c_int64, c_int, c_int]),
    "mprg_kmeans_fit_lds": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 7),
    "mprg_kmeans_select": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 5),
    "mprg_cluster_further": (c_int, [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_int] + [c_void_p] * 6),
    "mprg_cluster_further_bounded": (c_int, [c_void_p] * 4 + [c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p, c_int] + [c_void_p] * 5
                                     + [ctypes.c_longlong, c_void_p]),
    "mprg_cluster_loop": (c_int, [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 14 + [c_int, c_void_p]),
    "mprg_split_children": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 7),
    "mprg_leaf_jobs": (c_int, [c_void_p, c_int64] + [c_void_p] * 6),
    "mprg_emit_alleles": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "mprg_kmeans_speculative_kinfo": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, ctypes.c_longlong, c_void_p, c_void_p, c_void_p]),
    "mprg_export_alignments": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, ctypes.c_longlong, c_void_p, c_void_p]),
    "mprg_forest_level": (c_int, [c_void_p, c_void_p]),
    "mprg_forest_state_init": (c_int, [c_void_p, ctypes.c_longlong, ctypes.c_longlong, c_void_p]),
    "mprg_forest_state_rewind": (c_int, [c_void_p, ctypes.c_longlong, ctypes.c_longlong, c_void_p]),
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
    "mprg_ingest_open_mem_host": (c_void_p, [c_void_p, c_void_p, ctypes.c_longlong, c_int]),
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
    "mprg_rt_device_count": (c_int, []),
    "mprg_rt_init": (c_int, [c_int]),
    "mprg_rt_malloc": (c_void_p, [ctypes.c_longlong]),
    "mprg_rt_free": (c_int, [c_void_p]),
    "mprg_rt_host_malloc": (c_void_p, [ctypes.c_longlong]),
    "mprg_rt_host_free": (c_int, [c_void_p]),
    "mprg_rt_stream_create": (c_void_p, []),
    "mprg_rt_stream_destroy": (c_int, [c_void_p]),
    "mprg_rt_stream_sync": (c_int, [c_void_p]),
    "mprg_rt_memcpy_async": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p]),
    "mprg_rt_memset_async": (c_int, [c_void_p, c_int, ctypes.c_longlong, c_void_p]),
    "mprg_rt_event_create": (c_void_p, [c_int]),
    "mprg_rt_event_destroy": (c_int, [c_void_p]),
    "mprg_rt_event_record": (c_int, [c_void_p, c_void_p]),
    "mprg_rt_event_sync": (c_int, [c_void_p]),
    "mprg_rt_event_query": (c_int, [c_void_p]),
    "mprg_rt_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "mprg_rt_event_elapsed_ms": (ctypes.c_double, [c_void_p, c_void_p]),
}


class MprgError(RuntimeError):
    pass


def _library_is_mapped(path: str) -> bool:
    """Is this shared object already loaded in this process?"""
    try:
        real = os.path.realpath(path)
        with open("/proc/self/maps") as fh:
            return any(line.rstrip().endswith(real) for line in fh)
    except OSError:
        return False


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

    def call(self, name, *args, work: float = 0.0, side: Optional[int] = None, label: Optional[str] = None):
        """Enqueue one C-ABI entry point.  `work` = algorithmic bytes of this launch (roofline accounting).  side: the launch
        goes to side stream `side` (its stream argument must be side_ptr(side)): the timing events are recorded there.
        label: the name the launch is timed under (an entry point that launches different kernels by argument), default: name."""
        key = label or name
        ev = self._event_pair() if self.profile is not None and (self.profile_only is None or key in self.profile_only) else None
        if ev:
            self._record(ev[0], side)
        rc = getattr(self.lib, name)(*args)
        if ev:
            self._record(ev[1], side)
            self.profile.setdefault(key, []).append((ev[0], ev[1], float(work)))
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
        import sys
        if "torch" not in sys.modules and _library_is_mapped(lib_path):
            # torch loads ITS copy of the HIP runtime by path; a library bound to the system's copy would then hand torch's
            # pointers to a runtime that never saw them (hipMemsetAsync: invalid value)
            raise MprgError(f"{os.path.basename(lib_path)} was loaded in this process before torch: use the runtime backend "
                            "(MPRG_BACKEND=runtime, make_backend('runtime')) or import torch first")
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
        with self.torch.cuda.stream(self.stream_obj):          # (the caching allocator hands a freed block to later allocations of the SAME stream)
            return self.torch.empty(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)

    # (fills and copies are enqueued on the BACKEND's stream whatever torch's current stream is: a forest that is enqueued without
    #  host waits has nothing else that would order them against the kernels)
    def zeros(self, nbytes: int):
        with self.torch.cuda.stream(self.stream_obj):
            return self.torch.zeros(max(int(nbytes), 16), dtype=self.torch.uint8, device=self.device)

    def full(self, nbytes: int, byte: int):
        with self.torch.cuda.stream(self.stream_obj):
            return self.torch.full((max(int(nbytes), 16),), int(byte), dtype=self.torch.uint8, device=self.device)

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
        with self.torch.cuda.stream(self.stream_obj):          # (behind everything enqueued on the backend's stream)
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
        with self.torch.cuda.stream(self.stream_obj):
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
        with self.torch.cuda.stream(self.stream_obj):
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

    n_side_streams = 5

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


class _RtBuffer:
    """Device memory of HipRuntimeBackend: an address and a size.  Dropping the last reference hands the block back to the
    backend's free lists (work enqueued earlier on the backend's stream runs before anything a later owner enqueues there)."""
    __slots__ = ("mprg_addr", "nbytes", "_cap", "_owner", "__weakref__")

    def __init__(self, owner, addr: int, cap: int, nbytes: int):
        self._owner, self.mprg_addr, self._cap, self.nbytes = owner, addr, cap, nbytes

    def __len__(self):
        return self.nbytes

    def __del__(self):
        owner = self._owner
        if owner is not None and owner._free is not None:
            owner._free.setdefault(self._cap, []).append(self.mprg_addr)


class _RtHostBuffer:
    """Page-locked host memory of HipRuntimeBackend (freed with the backend: un-pinning is slow, these are few and reused)."""
    __slots__ = ("mprg_addr", "nbytes", "array")

    def __init__(self, addr: int, nbytes: int):
        self.mprg_addr, self.nbytes = addr, nbytes
        self.array = np.frombuffer((ctypes.c_ubyte * nbytes).from_address(addr), np.uint8)

    def numel(self):
        return self.nbytes


_rt_thread = __import__("threading").local()          # .device: the HIP device this THREAD was last switched to


class _RtEvent:
    def __init__(self, lib, timing: bool):
        self.lib = lib
        self.h = lib.mprg_rt_event_create(int(timing))
        if not self.h:
            raise MprgError(f"event: {lib.mprg_last_error().decode()}")

    def record(self, stream):
        self.lib.mprg_rt_event_record(self.h, stream)

    def synchronize(self):
        self.lib.mprg_rt_event_sync(self.h)

    def elapsed_time(self, other) -> float:
        other.synchronize()
        return self.lib.mprg_rt_event_elapsed_ms(self.h, other.h)

    def __del__(self):
        try:
            self.lib.mprg_rt_event_destroy(self.h)
        except Exception:
            pass


class HipRuntimeBackend(_Base):
    """The same product backend without torch: device memory, page-locked host memory, streams and events through the library's
    own mprg_rt_* calls (include/mprg.h), a size-class free list instead of torch's caching allocator.  The command line uses it
    (a run starts ~1 s earlier: tools/startup_probe.py); anything that also needs torch.distributed keeps HipBackend."""
    name = "hip (no torch)"

    def __init__(self, device: Optional[int] = None, lib_path: str = HIP_LIB_PATH, own_stream: bool = True):
        if not os.path.exists(lib_path):
            raise MprgError(f"{lib_path} not found: build it with `python __graft_entry__.py build` (hipcc, gfx950)")
        self._free = None
        self.lib = bind(ctypes.CDLL(lib_path))
        n = self.lib.mprg_rt_device_count()
        if n <= 0:
            raise MprgError("no ROCm device visible: make_prg_amd has no CPU fallback")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % n
        self.device = device
        _rt_thread.device = None
        self._on_device()
        self.stream = self._ptr(self.lib.mprg_rt_stream_create(), "stream")
        self._free = {}                       # capacity -> addresses ready for reuse
        self._host = []                       # page-locked blocks, freed at close()
        self._pending = []                    # (event, buffer): buffers the copy stream still reads
        self.n_cus = self.lib.mprg_device_cus()

    def _init_rings(self):
        if not hasattr(self, "_pinned"):
            self._pinned, self._parity, self._retired, self._turns = {}, {}, [], {}
            self._copy_stream = self._ptr(self.lib.mprg_rt_stream_create(), "stream")

    def clone(self):
        """A second backend on the same device: a compute stream, a copy stream, device free lists and a header block of its OWN —
        and THIS backend's page-locked host memory (the upload arenas by key, the download rings and their turn counters): two
        engines that take a pipeline's chunks in turn (pipeline.py) then cycle through one set of pinned buffers in chunk order,
        as one engine would, instead of page-locking a second set (0.2 s per GB).
        INVARIANT the caller keeps: a shared block is handed to one backend at a time.  `_host_release` (a block that regrows, a retired
        ring block) waits for the CALLING backend's streams only before it frees page-locked memory — not for the sibling's upload or
        download that may still use it.  pipeline.py guarantees it with its `slot_free` semaphore (chunk i - 3 is fully written before
        its buffers are taken again); any other user of clone() must serialise the two backends' use of a block the same way."""
        other = HipRuntimeBackend(self.device)
        self._on_device()
        self._init_rings()
        if not hasattr(self, "_pinned_up"):
            self._pinned_up = {}
        other._pinned, other._parity, other._retired, other._turns = self._pinned, self._parity, self._retired, self._turns
        other._pinned_up, other._host, other._owns_host = self._pinned_up, self._host, False
        other._copy_stream = other._ptr(other.lib.mprg_rt_stream_create(), "stream")
        if hasattr(self, "async_depth"):
            other.async_depth = self.async_depth
        return other

    def _check(self, rc, what):
        if rc != 0:
            raise MprgError(f"{what} failed ({rc}): {self.lib.mprg_last_error().decode()}")

    def _on_device(self):
        """HIP's current device is a property of the calling THREAD (a new thread starts on device 0): every allocation, stream,
        event and launch of this backend first makes sure the thread is on the backend's device — bench.py and the tests drive
        engines from pool threads, and ranks of a multi-GPU job do not sit on device 0."""
        if getattr(_rt_thread, "device", None) != self.device:
            self._check(self.lib.mprg_rt_init(self.device), "device")
            _rt_thread.device = self.device

    def call(self, name, *args, work: float = 0.0, side: Optional[int] = None, label: Optional[str] = None):
        self._on_device()
        return super().call(name, *args, work=work, side=side, label=label)

    def _ptr(self, p, what):
        if not p:
            raise MprgError(f"{what} failed: {self.lib.mprg_last_error().decode()}")
        return p

    def on_stream(self):
        import contextlib
        return contextlib.nullcontext()       # every entry point takes the stream explicitly

    @staticmethod
    def _capacity(nbytes: int) -> int:
        """Size classes 1/8 of a power of two apart (at most 12.5 % unused), 512-byte floor."""
        n = max(int(nbytes), 512)
        step = 1 << max(n.bit_length() - 4, 9)
        return (n + step - 1) // step * step

    def empty(self, nbytes: int):
        nbytes = max(int(nbytes), 16)
        cap = self._capacity(nbytes)
        free = self._free.get(cap)
        if free:
            return _RtBuffer(self, free.pop(), cap, nbytes)
        self._on_device()
        addr = self.lib.mprg_rt_malloc(cap)
        if not addr:          # give what the free lists hold back to the runtime and try once more
            self.trim()
            addr = self._ptr(self.lib.mprg_rt_malloc(cap), f"device allocation of {cap} bytes")
        return _RtBuffer(self, addr, cap, nbytes)

    def trim(self):
        """hipFree of every block in the free lists (after waiting for the stream)."""
        self.synchronize()
        for cap, addrs in self._free.items():
            for a in addrs:
                self.lib.mprg_rt_free(a)
        self._free = {}

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        if arr.nbytes == 0:
            return self.empty(16)
        buf = self.empty(arr.nbytes)
        self._on_device()
        self._check(self.lib.mprg_rt_memcpy_async(buf.mprg_addr, arr.ctypes.data, arr.nbytes, 1, self.stream), "upload")
        self._check(self.lib.mprg_rt_stream_sync(self.stream), "upload")          # pageable source: it may go away after this call
        return buf

    def download(self, buf, dtype, count: int) -> np.ndarray:
        out = np.empty(int(count), dtype)
        if out.nbytes:
            self._on_device()
            self._check(self.lib.mprg_rt_memcpy_async(out.ctypes.data, self.ptr(buf), out.nbytes, 2, self.stream), "download")
            self._check(self.lib.mprg_rt_stream_sync(self.stream), "download")
        return out

    def _host_block(self, nbytes: int) -> _RtHostBuffer:
        self._on_device()
        hb = _RtHostBuffer(self._ptr(self.lib.mprg_rt_host_malloc(int(nbytes)), f"page-locked allocation of {nbytes} bytes"), int(nbytes))
        self._host.append(hb)
        return hb

    def _host_release(self, hb: Optional[_RtHostBuffer]):
        """Un-pins a block that was REPLACED by a bigger one (pinned() / download_async regrowth).  Everything the streams hold
        is waited for first: a copy into or out of the old block may still be queued.  Rare (buffers grow by 1/8 headroom)."""
        if hb is None or hb not in self._host:
            return
        self.synchronize()
        if getattr(self, "_copy_stream", None):
            self._check(self.lib.mprg_rt_stream_sync(self._copy_stream), "synchronize")
        self._host.remove(hb)
        hb.array = None
        self.lib.mprg_rt_host_free(hb.mprg_addr)

    def close(self):
        """Gives everything back to the runtime: waits for the streams, then hipFree of the free lists, hipHostFree of every
        page-locked block, the streams destroyed.  Buffers still referenced by the caller must not be used afterwards (their
        blocks are NOT freed: they were never returned to the free lists).  Idempotent; also run when the backend is collected."""
        if self._free is None:
            return
        try:
            self._on_device()
            self.lib.mprg_rt_stream_sync(self.stream)
            for st in [getattr(self, "_copy_stream", None)] + list(getattr(self, "_side", [])):
                if st:
                    self.lib.mprg_rt_stream_sync(st)
            self._pending = []
            free, self._free = self._free, None          # (None: a buffer dropped from now on is not put on a list)
            for addrs in free.values():
                for a in addrs:
                    self.lib.mprg_rt_free(a)
            for hb in (self._host if getattr(self, "_owns_host", True) else []):          # (a clone() leaves the host memory to its origin)
                hb.array = None
                self.lib.mprg_rt_host_free(hb.mprg_addr)
            self._host = []
            self._pinned_up, self._pinned, self._hdr_block = {}, {}, None
            for st in [getattr(self, "_copy_stream", None)] + list(getattr(self, "_side", [])) + [self.stream]:
                if st:
                    self.lib.mprg_rt_stream_destroy(st)
            self._copy_stream, self._side, self.stream = None, [], None
        except Exception:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pinned(self, nbytes: int, key):
        if not hasattr(self, "_pinned_up"):
            self._pinned_up = {}
        hb = self._pinned_up.get(key)
        if hb is None or hb.nbytes < nbytes:
            self._host_release(hb)
            hb = self._pinned_up[key] = self._host_block(max(int(nbytes) + (int(nbytes) >> 3), 1 << 20))
        return hb, hb.array

    def upload_from(self, pinned_buf, nbytes: int):
        if nbytes == 0:
            return self.empty(16)
        buf = self.empty(nbytes)
        self._on_device()
        self._check(self.lib.mprg_rt_memcpy_async(buf.mprg_addr, pinned_buf.mprg_addr, int(nbytes), 1, self.stream), "upload")
        return buf

    def host_visible(self, nbytes: int):
        """One block per backend, reused by every engine (an engine reads its header right after the wait for its own step;
        engines of one backend run one after the other on its stream)."""
        hb = getattr(self, "_hdr_block", None)
        if hb is None or hb.nbytes < nbytes:
            hb = self._hdr_block = self._host_block(max(int(nbytes), 16))
            hb.array[:] = 0
        return hb, hb.array[:max(int(nbytes), 16)]

    def _reap(self):
        """Drops the buffers whose copies on the copy stream are done."""
        while self._pending and self.lib.mprg_rt_event_query(self._pending[0][0].h) == 0:
            self._pending.pop(0)

    def download_async(self, buf, nbytes: int, group: int = 0):
        """As HipBackend.download_async: copy stream, `async_depth` page-locked buffers per group used in turn."""
        nbytes = int(nbytes)
        self._on_device()
        self._init_rings()
        depth = getattr(self, "async_depth", 2)
        par = self._parity.get(group, 0) % depth
        self._parity[group] = par + 1
        turn = self._turns[group] = self._turns.get(group, 0) + 1          # calls of this group so far
        host = self._pinned.get((group, par))
        if host is None or host.nbytes < nbytes:
            for q in range(depth):
                if self._pinned.get((group, q)) is None or self._pinned[(group, q)].nbytes < nbytes:
                    # the replaced block may still be read by whoever received it (a result stays valid until the depth-th
                    # following call of its group): it is un-pinned once the ring has gone round
                    if self._pinned.get((group, q)) is not None:
                        self._retired.append((group, turn + depth, self._pinned[(group, q)]))
                    self._pinned[(group, q)] = self._host_block(max(nbytes + (nbytes >> 3), 1 << 20))
            host = self._pinned[(group, par)]
        if self._retired:
            due = [r for r in self._retired if r[0] == group and r[1] <= turn]
            if due:
                self._retired = [r for r in self._retired if r not in due]
                for _, _, hb in due:
                    self._host_release(hb)
        self._reap()
        ready = _RtEvent(self.lib, False)
        ready.record(self.stream)
        self._check(self.lib.mprg_rt_stream_wait_event(self._copy_stream, ready.h), "copy stream")
        if nbytes:
            self._check(self.lib.mprg_rt_memcpy_async(host.mprg_addr, self.ptr(buf), nbytes, 2, self._copy_stream), "download")
        done = _RtEvent(self.lib, False)
        done.record(self._copy_stream)
        self._pending.append((done, buf, ready))          # buf must not be handed out again before the copy ran
        return host.array[:nbytes], done.synchronize

    def ptr(self, buf) -> int:
        return buf.mprg_addr

    def grown(self, buf, used_bytes: int, new_bytes: int):
        new = self.empty(int(new_bytes))
        if used_bytes:
            self._on_device()
            self._check(self.lib.mprg_rt_memcpy_async(new.mprg_addr, self.ptr(buf), int(used_bytes), 3, self.stream), "copy")
        return new

    def synchronize(self):
        self._on_device()
        self._check(self.lib.mprg_rt_stream_sync(self.stream), "synchronize")

    def _event_pair(self):
        self._on_device()
        return (_RtEvent(self.lib, True), _RtEvent(self.lib, True))

    def _record(self, event, side):
        event.record(self.stream if side is None else self._side[side])

    n_side_streams = 5

    def _sides(self, n):
        self._on_device()
        if not hasattr(self, "_side"):
            self._side = []
        while len(self._side) < n:
            self._side.append(self._ptr(self.lib.mprg_rt_stream_create(), "stream"))
        return self._side[:n]

    def side_ptr(self, i: int):
        return self._sides(i + 1)[i]

    def fork(self, n: int):
        self._on_device()
        ev = _RtEvent(self.lib, False)
        ev.record(self.stream)
        for s_ in self._sides(n):
            self.lib.mprg_rt_stream_wait_event(s_, ev.h)

    def join(self, n: int):
        for s_ in self._sides(n):
            ev = _RtEvent(self.lib, False)
            ev.record(s_)
            self.lib.mprg_rt_stream_wait_event(self.stream, ev.h)

    def zeros(self, nbytes: int):
        return self.full(nbytes, 0)

    def full(self, nbytes: int, byte: int):
        buf = self.empty(nbytes)
        self._on_device()
        self._check(self.lib.mprg_rt_memset_async(buf.mprg_addr, int(byte), buf.nbytes, self.stream), "memset")
        return buf


def make_backend(kind: Optional[str] = None, device: Optional[int] = None, own_stream: bool = True):
    """A product backend by name: "torch" (HipBackend) or "runtime" (HipRuntimeBackend); None: MPRG_BACKEND, else "runtime"."""
    kind = kind or os.environ.get("MPRG_BACKEND") or "runtime"
    if kind not in ("torch", "runtime"):
        raise ValueError(f"backend: torch or runtime, not {kind!r}")
    return HipRuntimeBackend(device) if kind == "runtime" else HipBackend(device, own_stream=own_stream)

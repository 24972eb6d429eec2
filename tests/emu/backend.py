"""TEST-ONLY stand-in for make_prg_amd.backend.HipBackend.

Compiles make_prg_amd/csrc/mprg_api.hip UNCHANGED with g++ against tests/emu/include/hip/hip_runtime.h, a test-only
stand-in for the HIP runtime header that runs one workgroup at a time with its threads as fibers (barriers, ballots and
shuffles are rendezvous points); "device" buffers are NumPy arrays.  It lets the GPU-less build
container check the kernel LOGIC and the host engine against the oracle.  The product never imports this module;
`-m gpu` tests and everything under make_prg_amd/ use the HIP library only.
"""
import ctypes
import os
import subprocess

import numpy as np

from make_prg_amd.backend import _Base, bind

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SRC_DIR = os.path.join(ROOT, "make_prg_amd", "csrc")
LIB = os.path.join(HERE, "_build", "libmprg_emu.so")


def build_emu(force=False, defines=(), tag="") -> str:
    """`defines`: extra -D switches of test-only variants (e.g. MPRG_TEST_WEAK_HASH: every row hash collides, so the
    exact-comparison fallbacks run); `tag` names the variant's library."""
    lib = LIB.replace(".so", f"{tag}.so")
    srcs = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR)] + [os.path.join(ROOT, "include", "mprg.h"),
            os.path.join(HERE, "include", "hip", "hip_runtime.h")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < newest:
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O2", "-ffp-contract=off", "-mfma",
                               "-I", os.path.join(HERE, "include"), '-DMPRG_BUILD_TAG="cpu emulation, tests only"',
                               "-fPIC", "-shared", "-Wno-unused-function", "-Wno-attributes", "-Wno-unknown-pragmas"] +
                              [f"-D{d}" for d in defines] +
                              ["-pthread", os.path.join(SRC_DIR, "mprg_api.hip"), "-o", lib, "-lz"])
    return lib


class EmuBackend(_Base):
    name = "cpu-emulation(test-only)"

    def __init__(self, defines=(), tag=""):
        self.lib = bind(ctypes.CDLL(build_emu(defines=defines, tag=tag)))
        self.stream = None
        self.n_cus = 1

    def clone(self):
        """A second backend over the same library (pipeline.py interleaves two engines on a backend and its clone)."""
        import copy
        return copy.copy(self)

    def empty(self, nbytes):
        return np.full(max(int(nbytes), 16), 0xA5, np.uint8)   # poison: catches reads of unwritten scratch

    def zeros(self, nbytes):
        return np.zeros(max(int(nbytes), 16), np.uint8)

    def full(self, nbytes, byte):
        return np.full(max(int(nbytes), 16), int(byte), np.uint8)

    def upload(self, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        return a if a.size else self.empty(16)

    def download(self, buf, dtype, count):
        nbytes = int(count) * np.dtype(dtype).itemsize
        return buf[:nbytes].copy().view(dtype)

    def pinned(self, nbytes, key):
        a = np.zeros(max(int(nbytes), 16), np.uint8)
        return a, a

    def upload_from(self, pinned_buf, nbytes):
        return pinned_buf[:int(nbytes)].copy() if nbytes else self.empty(16)

    def host_visible(self, nbytes):
        a = np.zeros(int(nbytes), np.uint8)
        return a, a

    def download_async(self, buf, nbytes, group=0):
        return buf[:int(nbytes)].copy(), (lambda: None)

    def ptr(self, buf):
        return buf.mprg_addr if hasattr(buf, "mprg_addr") else buf.ctypes.data

    def grown(self, buf, used_bytes, new_bytes):
        new = np.full(int(new_bytes), 0xA5, np.uint8)
        new[:used_bytes] = buf[:used_bytes]
        return new

    def synchronize(self):
        pass

"""Host-side native helpers of libmprg (include/mprg.h, functions ending in _host): the one-pass .bin and .gfa encoders.
The library is the product's own shared object (make_prg_amd/_lib/libmprg_hip.so); these entry points touch no GPU, so they
are also usable before / without a device.  If the library cannot be loaded the callers keep their Python forms."""
import ctypes
import os
from typing import Optional

import numpy as np

_lib = None
_tried = False


def set_library(lib):
    """Tests inject the emulation build (same sources) here."""
    global _lib, _tried
    _lib, _tried = lib, True


def library():
    global _lib, _tried
    if not _tried:
        _tried = True
        from ..backend import HIP_LIB_PATH, bind
        if os.path.exists(HIP_LIB_PATH):
            try:
                _lib = bind(ctypes.CDLL(HIP_LIB_PATH))
            except (OSError, AttributeError):
                _lib = None
    return _lib


def prg_encode(prg) -> Optional[np.ndarray]:
    """uint32 stream of a PRG string (str / bytes), or None if the native one-pass encoder does not apply."""
    lib = library()
    if lib is None:
        return None
    data = prg.encode("ascii", "replace") if isinstance(prg, str) else bytes(prg)
    out = np.empty(max(len(data), 1), np.uint32)
    n = lib.mprg_prg_encode_host(data, len(data), out.ctypes.data)
    return out[:n] if n >= 0 else None


def gfa_text(prg) -> Optional[bytes]:
    """GFA1 text (bytes) of a PRG string, or None if the native one-pass builder does not apply."""
    lib = library()
    if lib is None:
        return None
    data = prg.encode("ascii", "replace") if isinstance(prg, str) else bytes(prg)
    for cap in (4096 + 3 * len(data), 256 + 48 * len(data)):
        buf = ctypes.create_string_buffer(cap)
        n = lib.mprg_gfa_text_host(data, len(data), buf, cap)
        if n != -4:                      # MPRG_OUT_TOO_SMALL: once more with the worst-case bound
            break
    return buf.raw[:n] if n >= 0 else None


def parse_fasta(text):
    """(matrix uint8 [rows, columns] upper-cased, titles list) of a FASTA alignment, or None if the native parser does not
    apply (library missing, or bytes it leaves to the Python parser).  Raises ValueError for sequences of unequal length."""
    lib = library()
    if lib is None:
        return None
    data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    n_rec, seq_len = ctypes.c_longlong(0), ctypes.c_longlong(0)
    rc = lib.mprg_fasta_scan_host(data, len(data), ctypes.byref(n_rec), ctypes.byref(seq_len))
    if rc == -5:
        raise ValueError("Sequences must all be the same length")
    if rc != 0:
        return None
    S, L = n_rec.value, seq_len.value
    matrix = np.empty((S, L), np.uint8)
    spans = np.empty((max(S, 1), 2), np.int64)
    lib.mprg_fasta_fill_host(data, len(data), matrix.ctypes.data, L, spans.ctypes.data)
    titles = [data[a:b].decode("ascii") for a, b in spans[:S].tolist()]
    return matrix, titles

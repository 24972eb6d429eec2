"""Host-side native helpers (include/mprg.h, functions ending in _host that touch no device): the one-pass .bin and .gfa
encoders and the FASTA parser.  They are bound from libmprg_host.so — the same sources as in libmprg_hip.so, built with
plain g++ WITHOUT the HIP runtime (csrc/mprg_host.cpp) — because this module is used by parent processes that are about to
fork GPU workers and by processes that have not imported torch yet: loading the HIP library there would bring a HIP
runtime into the process too early.  If the library is missing the callers keep their Python forms."""
import ctypes
import os
from typing import Optional

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB_PATH = os.environ.get("MPRG_HOST_LIB") or os.path.join(_PKG, "_lib", "libmprg_host.so")
_LL, _P = ctypes.c_longlong, ctypes.c_void_p
HOST_SIGNATURES = {
    "mprg_prg_encode_host": (_LL, [_P, _LL, _P]),
    "mprg_gfa_text_host": (_LL, [_P, _LL, _P, _LL]),
    "mprg_fasta_scan_host": (_LL, [_P, _LL, _P, _P]),
    "mprg_fasta_fill_host": (_LL, [_P, _LL, _P, _LL, _P]),
    "mprg_ingest_open_host": (_P, [_P, _LL, ctypes.c_int]),
    "mprg_ingest_open_mem_host": (_P, [_P, _P, _LL, ctypes.c_int]),
    "mprg_ingest_info_host": (None, [_P, _P]),
    "mprg_ingest_fill_host": (None, [_P, _P, _P, _P, _P, ctypes.c_int]),
    "mprg_ingest_text_host": (_LL, [_P, _LL, _P]),
    "mprg_ingest_close_host": (None, [_P]),
    "mprg_encode_sizes_host": (None, [_P, _P, _P, _LL, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P]),
    "mprg_encode_fill_host": (None, [_P, _P, _P, _LL, ctypes.c_int] + [_P] * 7),
    "mprg_crc32_members_host": (None, [_P, _P, _P, _LL, ctypes.c_int, _P]),
    "mprg_write_pieces_host": (ctypes.c_int, [ctypes.c_int, _P, _P, _P, _LL, ctypes.c_int]),
    "mprg_encode_pool_new_host": (_P, []),
    "mprg_encode_pool_reset_host": (None, [_P]),
    "mprg_encode_pool_free_host": (None, [_P]),
    "mprg_encode_pool_info_host": (None, [_P, _P]),
    "mprg_encode_batch_host": (ctypes.c_int, [_P, _P, _P, _P, _LL, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_P] * 5),
    "mprg_crc32_host": (ctypes.c_uint32, [ctypes.c_uint32, _P, _LL]),
}
_lib = None
_tried = False


def bind_host(lib):
    for name, (res, args) in HOST_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def set_library(lib):
    """Tests inject a library here (the emulation build exports the same functions); None: Python forms only."""
    global _lib, _tried
    _lib, _tried = lib, True


def library():
    global _lib, _tried
    if not _tried:
        _tried = True
        if os.path.exists(HOST_LIB_PATH):
            try:
                _lib = bind_host(ctypes.CDLL(HOST_LIB_PATH))
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

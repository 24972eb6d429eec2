"""Process-wide default device backend (lazy): the HIP library + GPU, or an explicit override for tests."""
import os
from typing import Optional

_backend = None


def set_backend(backend) -> None:
    """Install a backend object explicitly (tests inject tests/emu here; the product never calls this)."""
    global _backend
    _backend = backend


def get_backend(kind: Optional[str] = None):
    """kind: "runtime" (HipRuntimeBackend: buffers, streams and events from the library's own mprg_rt_* plumbing, no torch
    import; the default — a command-line run starts ~1 s earlier and the host side of a step is ~5 % cheaper) or "torch"
    (HipBackend: torch.cuda buffers and streams).  MPRG_BACKEND overrides both."""
    global _backend
    if _backend is None:
        kind = os.environ.get("MPRG_BACKEND") or kind or "runtime"
        if kind not in ("torch", "runtime"):
            raise ValueError(f"MPRG_BACKEND: torch or runtime, not {kind!r}")
        from . import backend as b
        # both raise MprgError without libmprg_hip.so + a ROCm device: no fallback
        _backend = b.HipRuntimeBackend() if kind == "runtime" else b.HipBackend()
    return _backend


def backend_is_set() -> bool:
    return _backend is not None

"""Process-wide default device backend (lazy): the HIP library + GPU, or an explicit override for tests."""
from typing import Optional

_backend = None


def set_backend(backend) -> None:
    """Install a backend object explicitly (tests inject tests/emu here; the product never calls this)."""
    global _backend
    _backend = backend


def get_backend():
    global _backend
    if _backend is None:
        from .backend import HipBackend
        _backend = HipBackend()          # raises MprgError without libmprg_hip.so + a ROCm device: no fallback
    return _backend


def backend_is_set() -> bool:
    return _backend is not None

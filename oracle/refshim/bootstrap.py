"""Make the unmodified reference importable in THIS container (never on the GPU box).

    import oracle.refshim.bootstrap as rb; rb.install()
    import make_prg   # the real reference from /root/reference

Pins the oracle configuration of SURVEY.md §0.4/§0.6: scikit-learn KMeans with n_init=10 forced at the
reference's call site, OMP_NUM_THREADS=1, and an explicit OPENBLAS_CORETYPE (must be set before numpy loads).
"""
import importlib.metadata as _md
import os
import sys

REFERENCE_ROOT = "/root/reference"
PINNED_CORETYPE = "Haswell"
HERE = os.path.dirname(os.path.abspath(__file__))


def preset_env(coretype=PINNED_CORETYPE):
    """Must run before numpy/scipy are imported."""
    if "numpy" in sys.modules and os.environ.get("OPENBLAS_CORETYPE") != coretype:
        raise RuntimeError("numpy already imported with a different OPENBLAS_CORETYPE")
    os.environ["OPENBLAS_CORETYPE"] = coretype
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ["OPENBLAS_NUM_THREADS"] = "1"


def install(n_init=10):
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present (this only works in the build container)")
    if HERE not in sys.path:
        sys.path.insert(0, HERE)  # Bio, loguru stubs
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(1, REFERENCE_ROOT)
    _orig_version = _md.version

    def _version(name):
        if name == "make_prg":
            return "0.5.0"
        return _orig_version(name)

    _md.version = _version
    # intervaltree: pure python, only present in the conda py3.9 tree
    try:
        import intervaltree  # noqa
    except ImportError:
        sys.path.append("/opt/conda/lib/python3.9/site-packages/intervaltree/..")
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "intervaltree", "/opt/conda/lib/python3.9/site-packages/intervaltree/__init__.py",
            submodule_search_locations=["/opt/conda/lib/python3.9/site-packages/intervaltree"])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["intervaltree"] = mod
        spec.loader.exec_module(mod)
        sys.path.remove("/opt/conda/lib/python3.9/site-packages/intervaltree/..")
    import make_prg.from_msa.cluster_sequences as cs
    from sklearn.cluster import KMeans as _KMeans

    if n_init is not None:
        def KMeans(*a, **k):  # same call signature as the reference's call site
            k.setdefault("n_init", n_init)
            return _KMeans(*a, **k)
        cs.KMeans = KMeans
    return cs

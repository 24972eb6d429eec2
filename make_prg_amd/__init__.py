"""make_prg_amd — MI355X-native `from_msa` PRG construction with make_prg's Python surface for that path."""
__version__ = "0.1.0"
__all__ = ["MSA", "from_msa", "subcommands"]

from .msa import MSA, Record  # noqa: E402,F401

"""Minimal Biopython stand-in used ONLY inside the build container to import the
unmodified reference from /root/reference (Biopython is not installed here).
Test/golden-generation tooling: never shipped, never imported by the product."""
from . import Seq as _Seq, SeqRecord as _SeqRecord, Align, AlignIO, SeqIO, pairwise2  # noqa

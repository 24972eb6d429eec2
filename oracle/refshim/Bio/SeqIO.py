from .AlignIO import _parse_fasta
from .Seq import Seq
from .SeqRecord import SeqRecord, fasta_text


def write(records, handle, fmt):
    assert fmt == "fasta"
    if isinstance(records, SeqRecord):
        records = [records]
    close = False
    if isinstance(handle, (str, bytes)) or hasattr(handle, "__fspath__"):
        handle, close = open(handle, "w"), True
    n = 0
    try:
        for r in records:
            handle.write(fasta_text(r))
            n += 1
    finally:
        if close:
            handle.close()
    return n


def parse(handle, fmt):
    assert fmt == "fasta"
    close = False
    if isinstance(handle, (str, bytes)) or hasattr(handle, "__fspath__"):
        handle, close = open(handle), True
    try:
        for title, seq in list(_parse_fasta(handle)):
            w = title.split(None, 1)[0] if title.split() else ""
            yield SeqRecord(Seq(seq), id=w, name=w, description=title)
    finally:
        if close:
            handle.close()

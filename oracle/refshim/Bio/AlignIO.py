from .Align import MultipleSeqAlignment
from .Seq import Seq
from .SeqRecord import SeqRecord


def _parse_fasta(handle):
    title, chunks = None, []
    for line in handle:
        if line.startswith(">"):
            if title is not None:
                yield title, "".join(chunks)
            title, chunks = line[1:].rstrip(), []
        elif title is not None:
            chunks.append("".join(line.split()))
    if title is not None:
        yield title, "".join(chunks)


def read(handle, fmt):
    assert fmt == "fasta"
    close = False
    if isinstance(handle, str):
        handle, close = open(handle), True
    try:
        records = []
        for title, seq in _parse_fasta(handle):
            try:
                first_word = title.split(None, 1)[0]
            except IndexError:
                first_word = ""
            records.append(SeqRecord(Seq(seq), id=first_word, name=first_word, description=title))
    finally:
        if close:
            handle.close()
    if not records:
        raise ValueError("No records found in handle")
    return MultipleSeqAlignment(records)

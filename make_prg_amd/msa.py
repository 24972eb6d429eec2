"""Alignment container of the MI355X path.

The reference's boundary type is `make_prg.MSA = Bio.AlignIO.MultipleSeqAlignment` (make_prg/__init__.py:7-9).
This module provides the part of that contract the from_msa path uses: len(), iteration over
records with .id/.seq/.description, get_alignment_length(), msa[i], msa[:, a:b], MSA(records),
format(msa, "fasta") — backed by one uint8 rows x columns matrix (ASCII bytes on the host; 1-byte symbol codes on
the device, see include/mprg.h), so that slicing is a view and packing for the GPU is one table lookup.
"""
import gzip
import hashlib
import random
from collections import Counter
from io import StringIO
from typing import Iterable, List, Optional, Sequence, Union

import numpy as np

ALPHABET = b"ACGT-RYKMSWN"
CODE_GAP, CODE_N = 4, 11
_ENC = np.full(256, 255, np.uint8)
for _i, _b in enumerate(ALPHABET):
    _ENC[_b] = _i
_DEC = np.frombuffer(ALPHABET + b"?" * (256 - len(ALPHABET)), dtype=np.uint8)


def encode(ascii_matrix: np.ndarray) -> np.ndarray:
    """ASCII bytes -> device symbol codes; 255 marks a byte outside ACGT-RYKMSWN."""
    return _ENC[ascii_matrix]


def decode(codes: np.ndarray) -> np.ndarray:
    return _DEC[codes]


class Record:
    """One row: what the path reads from a Bio.SeqRecord (id, seq, description)."""
    __slots__ = ("id", "description", "_data", "name")

    def __init__(self, seq: Union[str, bytes, np.ndarray], id: str = "<unknown id>", description: str = "<unknown description>",
                 name: Optional[str] = None):
        if isinstance(seq, np.ndarray):
            self._data = seq
        else:
            self._data = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
        self.id = id
        self.description = description
        self.name = id if name is None else name

    @property
    def seq(self) -> str:
        return self._data.tobytes().decode()

    def __len__(self):
        return int(self._data.shape[0])

    def __iter__(self):
        return iter(self.seq)

    def __getitem__(self, item):
        if isinstance(item, slice):
            return Record(self._data[item], self.id, self.description, self.name)
        return chr(self._data[item])


def _fasta_title(rid: str, desc: str) -> str:
    # Bio.SeqIO FastaWriter title rule
    if desc and desc.split(None, 1)[0] == rid:
        return desc
    if desc:
        return f"{rid} {desc}"
    return rid


class MSA:
    def __init__(self, records: Iterable[Record] = (), *, _data=None, _ids=None, _descs=None):
        self.pending_n = False          # True: holds N that the batch engine still has to replace (load_alignment_text)
        if _data is not None:
            self.data, self.ids, self.descriptions = _data, list(_ids), list(_descs)
            return
        records = list(records)
        if records:
            lengths = {len(r) for r in records}
            if len(lengths) != 1:
                raise ValueError("Sequences must all be the same length")
            self.data = np.stack([r._data for r in records]).astype(np.uint8, copy=False)
        else:
            self.data = np.zeros((0, 0), np.uint8)
        self.ids = [r.id for r in records]
        self.descriptions = [r.description for r in records]

    @classmethod
    def from_strings(cls, seqs: Sequence[str], ids: Optional[Sequence[str]] = None, descriptions=None) -> "MSA":
        ids = [f"s{i}" for i in range(len(seqs))] if ids is None else list(ids)
        descs = [""] * len(seqs) if descriptions is None else list(descriptions)
        return cls([Record(s, i, d) for s, i, d in zip(seqs, ids, descs)])

    def __len__(self):
        return int(self.data.shape[0])

    def get_alignment_length(self) -> int:
        return int(self.data.shape[1]) if len(self) else 0

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __getitem__(self, index):
        if isinstance(index, (int, np.integer)):
            return Record(self.data[index], self.ids[index], self.descriptions[index])
        if isinstance(index, slice):
            return MSA(_data=self.data[index], _ids=self.ids[index], _descs=self.descriptions[index])
        rows, cols = index
        if isinstance(rows, (int, np.integer)):
            return self[rows][cols]
        sub = self.data[rows]
        if isinstance(cols, (int, np.integer)):
            return sub[:, cols].tobytes().decode()
        return MSA(_data=sub[:, cols], _ids=self.ids[rows], _descs=self.descriptions[rows])

    def rows_as_strings(self) -> List[str]:
        return [self.data[i].tobytes().decode() for i in range(len(self))]

    def __format__(self, fmt):
        if fmt != "fasta":
            raise ValueError(f"unsupported format {fmt}")
        out = []
        for i in range(len(self)):
            out.append(f">{_fasta_title(self.ids[i], self.descriptions[i])}\n")
            s = self.data[i].tobytes().decode()
            out.extend(s[p:p + 60] + "\n" for p in range(0, len(s), 60))
        return "".join(out)

    def format(self, fmt):
        return self.__format__(fmt)


# ---------------------------------------------------------------------------------------------------- ingest (A0)
def _parse_fasta(text: str):
    title, chunks = None, []
    for line in text.splitlines():
        if line.startswith(">"):
            if title is not None:
                yield title, "".join(chunks)
            title, chunks = line[1:].rstrip(), []
        elif title is not None:
            chunks.append("".join(line.split()))
    if title is not None:
        yield title, "".join(chunks)


def read_fasta_alignment(text: str) -> MSA:
    """AlignIO.read(handle, "fasta") behaviour the path relies on (utils/io_utils.py:17-31)."""
    recs = []
    for title, seq in _parse_fasta(text):
        words = title.split(None, 1)
        recs.append(Record(seq, words[0] if words else "", title))
    if not recs:
        raise ValueError("No records found in handle")
    return MSA(recs)


def _majority_consensus(upper: np.ndarray, upto=None) -> np.ndarray:
    """utils/seq_utils.py:246-290, counts array-at-a-time: per symbol the column counts and the first row it occurs in
    (the reference's Counter lists tied residues in first-seen order); only the random.Random draws — one per column,
    each consuming a data-dependent number of bits — stay a loop.  Columns >= upto are not needed by the caller."""
    rng = random.Random()
    rng.seed(hashlib.sha256(np.ascontiguousarray(upper).tobytes()).digest())
    S, C = upper.shape
    n = C if upto is None else min(C, upto)
    out = np.full(C, ord("A"), np.uint8)
    if n == 0 or S == 0:
        for c in range(n):
            out[c] = ord(rng.choice("ACGT"))
        return out
    sub = upper[:, :n]
    syms = [int(x) for x in np.nonzero(np.bincount(sub.reshape(-1), minlength=256))[0] if x not in (ord("-"), ord("N"))]
    if not syms:
        for c in range(n):
            out[c] = ord(rng.choice("ACGT"))
        return out
    counts = np.empty((len(syms), n), np.int64)
    first = np.empty((len(syms), n), np.int64)
    for k, sym in enumerate(syms):
        hit = sub == sym
        counts[k] = hit.sum(axis=0)
        first[k] = hit.argmax(axis=0)
    top = counts.max(axis=0)
    tied = counts == top
    n_tied = tied.sum(axis=0)
    single = tied.argmax(axis=0).tolist()                             # the only candidate where n_tied == 1
    chars = [chr(x) for x in syms]
    one = [[ch] for ch in chars]
    top_l, n_tied_l, choice, res = top.tolist(), n_tied.tolist(), rng.choice, [65] * n
    for c in range(n):
        if top_l[c] == 0:
            res[c] = ord(choice("ACGT"))
        elif n_tied_l[c] == 1:
            res[c] = ord(choice(one[single[c]]))
        else:
            ks = np.nonzero(tied[:, c])[0]
            ks = ks[np.argsort(first[ks, c], kind="stable")]
            res[c] = ord(choice([chars[k] for k in ks]))
    out[:n] = res
    return out


def _majority_consensus_by_column(upper: np.ndarray, upto=None) -> np.ndarray:
    """Plain restatement, column by column with a Counter (tests compare _majority_consensus with it).
    utils/seq_utils.py:246-290: one Random(sha256(rows)) draw per column; choice among the most frequent
    non-gap non-N residues in first-seen order, or among ACGT if the column has none.  Columns >= upto are not needed
    by the caller (no N at or after them) and are left as 'A'."""
    rng = random.Random()
    rng.seed(hashlib.sha256(np.ascontiguousarray(upper).tobytes()).digest())
    S, C = upper.shape
    out = np.full(C, ord("A"), np.uint8)
    gap, n = ord("-"), ord("N")
    const_col = (upper == upper[0:1]).all(axis=0)
    for c in range(C if upto is None else min(C, upto)):
        col = upper[:, c]
        if const_col[c] and col[0] != gap and col[0] != n:
            out[c] = ord(rng.choice([chr(col[0])]))
            continue
        counts = Counter(chr(x) for x in col.tolist() if x != gap and x != n)
        if not counts:
            out[c] = ord(rng.choice("ACGT"))
            continue
        top = max(counts.values())
        out[c] = ord(rng.choice([r for r, k in counts.items() if k == top]))
    return out


def consensus_from_counts(upper: np.ndarray, counts: np.ndarray, first: np.ndarray, upto: int) -> np.ndarray:
    """The seeded choice of utils/seq_utils.py:246-290 given the per-column residue counts (10 residues "ACGTRYKMSW";
    counts[c, q], first[c, q] = first row holding residue q or a huge value) — what mprg_column_residue_counts produces on
    the device, or _residue_counts_host below.  One random.Random(sha256(rows)) draw per column 0..upto-1, in order."""
    rng = random.Random()
    rng.seed(hashlib.sha256(np.ascontiguousarray(upper).tobytes()).digest())
    C = upper.shape[1]
    out = np.full(C, ord("A"), np.uint8)
    residues = "ACGTRYKMSW"
    top = counts.max(axis=1) if counts.size else np.zeros(C, np.int64)
    choice = rng.choice
    for c in range(min(C, upto)):
        if top[c] == 0:
            out[c] = ord(choice("ACGT"))
            continue
        tied = np.nonzero(counts[c] == top[c])[0]
        if len(tied) > 1:
            tied = tied[np.argsort(first[c, tied], kind="stable")]
        out[c] = ord(choice([residues[q] for q in tied]))
    return out


def load_alignment_text(text: str, defer_n: bool = False, alignment_format: str = "fasta") -> MSA:
    """utils/io_utils.py:17-49: parse, upper-case, overwrite every N with its column's majority-consensus base.
    defer_n: leave the N in place and mark the alignment (`pending_n`): the batch engine then gets the column counts
    from the device (mprg_column_residue_counts) for all such alignments of a batch at once.
    alignment_format: "fasta", or one of utils/align_formats.FORMATS (restated readers, see there)."""
    from .utils import native
    if alignment_format.lower() != "fasta":
        from .utils.align_formats import read_alignment
        recs = read_alignment(text, alignment_format)
        msa = MSA([Record(seq, rid, desc) for rid, desc, seq in recs])
        data = msa.data.copy()
        lower = (data >= ord("a")) & (data <= ord("z"))
        data[lower] -= 32
        out = MSA(_data=data, _ids=msa.ids, _descs=msa.descriptions)
        parsed = None
    else:
        parsed = native.parse_fasta(text)          # libmprg's two-pass host parser (plain ASCII text; else the Python one)
    if alignment_format.lower() != "fasta":
        pass
    elif parsed is not None:
        data, titles = parsed
        if len(titles) == 0:
            raise ValueError("No records found in handle")
        ids = []
        for title in titles:
            words = title.split(None, 1)
            ids.append(words[0] if words else "")
        out = MSA(_data=data, _ids=ids, _descs=titles)
    else:
        msa = read_fasta_alignment(text)
        data = msa.data.copy()
        lower = (data >= ord("a")) & (data <= ord("z"))
        data[lower] -= 32
        out = MSA(_data=data, _ids=msa.ids, _descs=msa.descriptions)
    is_n = data == ord("N")
    if is_n.any():          # the consensus is only ever used to overwrite N (io_utils.py:36-47): no N, no work
        if defer_n:
            out.pending_n = True
            return out
        last = int(np.nonzero(is_n.any(axis=0))[0].max())
        cons = _majority_consensus(data, upto=last + 1)     # the RNG stream must still advance column by column
        data[is_n] = np.broadcast_to(cons, data.shape)[is_n]
    return out


def load_alignment_file(msa_file, alignment_format: str = "fasta", defer_n: bool = False) -> MSA:
    if isinstance(msa_file, StringIO):
        return load_alignment_text(msa_file.getvalue(), defer_n, alignment_format)
    path = str(msa_file)
    if path.endswith(".gz"):
        with gzip.open(path, "rt") as fh:
            return load_alignment_text(fh.read(), defer_n, alignment_format)
    with open(path) as fh:
        return load_alignment_text(fh.read(), defer_n, alignment_format)

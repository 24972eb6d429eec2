from .SeqRecord import SeqRecord, fasta_text


class MultipleSeqAlignment:
    def __init__(self, records=(), **_):
        self._records = []
        for r in records:
            self.append(r)

    def append(self, record):
        if self._records and len(record) != self.get_alignment_length():
            raise ValueError("Sequences must all be the same length")
        self._records.append(record)

    def extend(self, records):
        for r in records:
            self.append(r)

    def get_alignment_length(self):
        return len(self._records[0]) if self._records else 0

    def __len__(self):
        return len(self._records)

    def __iter__(self):
        return iter(self._records)

    def __getitem__(self, index):
        if isinstance(index, int):
            return self._records[index]
        if isinstance(index, slice):
            return MultipleSeqAlignment(self._records[index])
        row_index, col_index = index
        if isinstance(row_index, int):
            return self._records[row_index][col_index]
        if isinstance(col_index, int):
            return "".join(rec[col_index] for rec in self._records[row_index])
        return MultipleSeqAlignment(rec[col_index] for rec in self._records[row_index])

    def format(self, fmt):
        return self.__format__(fmt)

    def __format__(self, fmt):
        assert fmt == "fasta"
        return "".join(fasta_text(r) for r in self._records)

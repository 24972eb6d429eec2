from .Seq import Seq


class SeqRecord:
    def __init__(self, seq, id="<unknown id>", name="<unknown name>",
                 description="<unknown description>", **_):
        self.seq = seq
        self.id = id
        self.name = name
        self.description = description

    def __len__(self):
        return len(self.seq)

    def __iter__(self):
        return iter(self.seq)

    def __getitem__(self, item):
        if isinstance(item, slice):
            return SeqRecord(Seq(str(self.seq)[item]), id=self.id, name=self.name,
                             description=self.description)
        return self.seq[item]

    def format(self, fmt):
        return self.__format__(fmt)

    def __format__(self, fmt):
        assert fmt == "fasta"
        return fasta_text(self)


def fasta_text(record, wrap=60):
    rid = str(record.id)
    desc = str(record.description)
    if desc and desc.split(None, 1)[0] == rid:
        title = desc
    elif desc:
        title = f"{rid} {desc}"
    else:
        title = rid
    data = str(record.seq)
    lines = [f">{title}\n"]
    for i in range(0, len(data), wrap):
        lines.append(data[i:i + wrap] + "\n")
    return "".join(lines)

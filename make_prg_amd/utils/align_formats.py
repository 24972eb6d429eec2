"""Alignment formats other than FASTA for `-f` (reference utils/io_utils.py:17-31: AlignIO.read(handle, alignment_format) takes
any Biopython AlignIO input format).  Restated from the format definitions and from what Biopython 1.79's readers keep of a
file — the record order (first appearance), the record ids, the concatenation of interleaved blocks — for the formats below.
Biopython is not installed in the build container and the reference's own tests hold FASTA only, so these readers are NOT
pinned against the reference (tests/test_align_formats.py checks them against the FASTA reader on hand-written files);
formats not listed raise ValueError naming the limitation."""
from typing import List, Tuple

Records = List[Tuple[str, str, str]]          # (id, description, sequence)
FORMATS = ("clustal", "stockholm", "phylip", "phylip-sequential", "phylip-relaxed")
_CLUSTAL_HEADERS = ("CLUSTAL", "PROBCONS", "MUSCLE", "MSAPROBS", "Kalign")


def read_alignment(text: str, fmt: str) -> Records:
    """The records of the ONE alignment in `text` (AlignIO.read: none -> "No records found in handle", a second one ->
    "More than one record found in handle")."""
    fmt = fmt.lower()
    if fmt == "clustal":
        alignments = _clustal(text)
    elif fmt == "stockholm":
        alignments = _stockholm(text)
    elif fmt in ("phylip", "phylip-relaxed"):
        alignments = _phylip(text, relaxed=fmt == "phylip-relaxed", sequential=False)
    elif fmt == "phylip-sequential":
        alignments = _phylip(text, relaxed=False, sequential=True)
    else:
        raise ValueError(f"alignment format {fmt!r} is not supported by the MI355X path: fasta, {', '.join(FORMATS)} are "
                         "(the reference takes any Biopython AlignIO format)")
    if not alignments:
        raise ValueError("No records found in handle")
    if len(alignments) > 1:
        raise ValueError("More than one record found in handle")
    records = alignments[0]
    if len({len(s) for _, _, s in records}) > 1:
        raise ValueError("Sequences must all be the same length")
    return records


def _clustal(text: str) -> List[Records]:
    """CLUSTAL W / X (.aln): a header line, then blocks of `id  residues [running count]` lines, each block followed by an
    optional conservation line (starts with a blank) and blank lines; the blocks' rows repeat the ids in the same order."""
    lines = text.splitlines()
    i = 0
    while i < len(lines) and not lines[i].strip():
        i += 1
    if i == len(lines):
        return []
    alignments: List[Records] = []
    while i < len(lines):
        head = lines[i]
        if not head.startswith(_CLUSTAL_HEADERS):
            raise ValueError(f"{head.split()[0] if head.split() else head!r} is not a known CLUSTAL header: {', '.join(_CLUSTAL_HEADERS)}")
        i += 1
        ids: List[str] = []
        seqs: List[List[str]] = []
        row = 0
        first_block = True
        in_block = False
        while i < len(lines) and not lines[i].startswith(_CLUSTAL_HEADERS):
            line = lines[i]
            i += 1
            if not line.strip() or line[0] in " \t":          # blank or conservation line: a block ends
                if in_block:
                    if not first_block and row != len(ids):
                        raise ValueError("a block of the CLUSTAL file has fewer rows than the first")
                    first_block, in_block, row = False, False, 0
                continue
            fields = line.split()
            if len(fields) < 2 or len(fields) > 3:
                raise ValueError(f"could not parse line:\n{line}")
            in_block = True
            if first_block:
                ids.append(fields[0])
                seqs.append([fields[1]])
            else:
                if row >= len(ids) or fields[0] != ids[row]:
                    raise ValueError(f"identifiers out of order? got {fields[0]!r}")
                seqs[row].append(fields[1])
            if len(fields) == 3:
                try:
                    letters = int(fields[2])
                except ValueError:
                    raise ValueError(f"could not parse line, bad sequence number:\n{line}") from None
                got = len("".join(seqs[row if not first_block else -1]).replace("-", ""))
                if got != letters:
                    raise ValueError(f"could not parse line, invalid sequence number:\n{line}")
            row += 1
        if ids:
            alignments.append([(n, n, "".join(parts)) for n, parts in zip(ids, seqs)])
    return alignments


def _stockholm(text: str) -> List[Records]:
    """Stockholm 1.0 (Pfam / Rfam): `# STOCKHOLM 1.0`, `#=G?` mark-up lines (ignored here: the path reads ids and residues),
    `id  residues` lines — an id may come back in later blocks, its pieces are joined — and `//` ends an alignment."""
    alignments: List[Records] = []
    lines = iter(text.splitlines())
    for line in lines:
        if not line.strip():
            continue
        if line.strip() != "# STOCKHOLM 1.0":
            raise ValueError("Did not find STOCKHOLM header")
        ids: List[str] = []
        seqs = {}
        closed = False
        for line in lines:
            line = line.strip()
            if line == "//":
                closed = True
                break
            if not line or line.startswith("#"):
                continue
            parts = [x.strip() for x in line.split(" ", 1)]
            if len(parts) != 2:
                raise ValueError(f"Could not split line into identifier and sequence:\n{line}")
            name, piece = parts[0], parts[1].replace(" ", "")
            if name not in seqs:
                ids.append(name)
                seqs[name] = []
            seqs[name].append(piece)
        if ids:
            alignments.append([(n, n, "".join(seqs[n])) for n in ids])
        elif closed:
            raise ValueError("No sequences found in the STOCKHOLM alignment")
    return alignments


def _phylip(text: str, relaxed: bool, sequential: bool) -> List[Records]:
    """PHYLIP: `<taxa> <sites>`, then one line per taxon: the name — the first 10 characters (strict) or the first word
    (relaxed) — and residues (blanks allowed); interleaved files continue in further blocks of one line per taxon without
    names; sequential files give each taxon all its lines before the next taxon starts."""
    lines = [l.rstrip("\n") for l in text.splitlines()]
    i = 0
    alignments: List[Records] = []

    def is_header(l: str) -> bool:
        p = l.split()
        return len(p) == 2 and p[0].isdigit() and p[1].isdigit()

    while i < len(lines):
        if not lines[i].strip():
            i += 1
            continue
        if not is_header(lines[i]):
            raise ValueError("First line should have two integers")
        n_taxa, n_sites = (int(x) for x in lines[i].split())
        i += 1
        ids: List[str] = []
        seqs: List[str] = []

        def split_name(l: str):
            if relaxed:
                p = l.split(None, 1)
                if len(p) != 2:
                    raise ValueError(f"could not split a PHYLIP line into name and sequence:\n{l}")
                return p[0], p[1].strip().replace(" ", "")
            return l[:10].strip(), l[10:].strip().replace(" ", "")

        if sequential:
            for _ in range(n_taxa):
                while i < len(lines) and not lines[i].strip():
                    i += 1
                if i == len(lines):
                    raise ValueError("the PHYLIP file ends before all taxa were read")
                name, s = split_name(lines[i])
                i += 1
                while len(s) < n_sites:
                    if i == len(lines):
                        raise ValueError(f"sequence {name} is shorter than the {n_sites} sites of the header")
                    s += lines[i].strip().replace(" ", "")
                    i += 1
                if len(s) > n_sites:
                    raise ValueError(f"sequence {name} is longer than the {n_sites} sites of the header")
                ids.append(name)
                seqs.append(s)
        else:
            for _ in range(n_taxa):
                if i == len(lines) or not lines[i].strip():
                    raise ValueError("the PHYLIP file ends before all taxa were read")
                name, s = split_name(lines[i])
                i += 1
                ids.append(name)
                seqs.append(s)
            while i < len(lines):          # further blocks of the interleaved form
                while i < len(lines) and not lines[i].strip():
                    i += 1
                if i == len(lines) or is_header(lines[i]) and all(len(s) == n_sites for s in seqs):
                    break
                for r in range(n_taxa):
                    if i == len(lines) or not lines[i].strip():
                        raise ValueError("a block of the interleaved PHYLIP file has fewer lines than taxa")
                    seqs[r] += lines[i].strip().replace(" ", "")
                    i += 1
            if any(len(s) != n_sites for s in seqs):
                raise ValueError(f"a sequence does not have the {n_sites} sites of the header")
        alignments.append([(n, n, s) for n, s in zip(ids, seqs)])
    return alignments

"""Structure checks that hold for the PRG of ANY recursion tree (size-independent properties for inputs too big for the oracle):
site markers nest, and every input row is spelt by exactly one choice of alleles.

PRG grammar (reference prg_builder.py:100-119, recursion_tree.py:222-300): text = item*; item = bases | site;
site = " s " allele (" s+1 " allele)* " s " with s odd >= 5, every site number used once; allele = item*."""
import re
from collections import Counter

_TOKEN = re.compile(r" (\d+) ")


class Site:
    __slots__ = ("number", "alleles", "pure")

    def __init__(self, number):
        self.number, self.alleles, self.pure = number, [[]], None


def parse_prg(prg: str):
    """PRG text -> nested items (str | Site); raises AssertionError if the markers do not nest."""
    parts = _TOKEN.split(prg)                  # literal, marker, literal, marker, ..., literal
    root = []
    stack = []                                  # open sites
    cur = root
    seen = set()
    for i, tok in enumerate(parts):
        if i % 2 == 0:
            if tok:
                assert set(tok) <= set("ACGT"), f"unexpected characters in {tok[:40]!r}"
                cur.append(tok)
            continue
        m = int(tok)
        assert m >= 5, "site markers start at 5"
        if m % 2 == 0:
            assert stack and stack[-1].number == m - 1, f"separator {m} outside its site"
            stack[-1].alleles.append([])
            cur = stack[-1].alleles[-1]
        elif stack and stack[-1].number == m:
            site = stack.pop()
            assert len(site.alleles) >= 2, f"site {m} has a single allele"
            cur = stack[-1].alleles[-1] if stack else root
        else:
            assert m not in seen, f"site {m} opens twice (or closes out of order)"
            seen.add(m)
            site = Site(m)
            cur.append(site)
            stack.append(site)
            cur = site.alleles[0]
    assert not stack, f"site {stack[-1].number} never closes" if stack else ""
    return root


def _prepare(items):
    for it in items:
        if isinstance(it, Site):
            if all(len(a) <= 1 and all(isinstance(x, str) for x in a) for a in it.alleles):
                by_len = {}
                for a in it.alleles:
                    s = a[0] if a else ""
                    by_len.setdefault(len(s), Counter())[s] += 1
                it.pure = by_len
            else:
                for a in it.alleles:
                    _prepare(a)


def _ends(items, row, pos, memo):
    """{end position: number of allele choices} of spelling row[pos:end] with `items`."""
    cur = {pos: 1}
    for it in items:
        nxt = {}
        if isinstance(it, str):
            n = len(it)
            for p, w in cur.items():
                if row.startswith(it, p):
                    nxt[p + n] = nxt.get(p + n, 0) + w
        else:
            for p, w in cur.items():
                key = (id(it), p)
                if key not in memo:
                    out = {}
                    if it.pure is not None:
                        for ln, strings in it.pure.items():
                            c = strings.get(row[p:p + ln], 0) if p + ln <= len(row) else 0
                            if c:
                                out[p + ln] = out.get(p + ln, 0) + c
                    else:
                        for a in it.alleles:
                            for e, c in _ends(a, row, p, memo).items():
                                out[e] = out.get(e, 0) + c
                    memo[key] = out
                for e, c in memo[key].items():
                    nxt[e] = nxt.get(e, 0) + w * c
        cur = nxt
        if not cur:
            break
    return cur


def spellings(tree, row: str) -> int:
    """Number of allele choices through the PRG that spell exactly `row`."""
    return _ends(tree, row, 0, {}).get(len(row), 0)


def check_prg_spells_rows(prg: str, rows, sample=None):
    """Markers nest, and every (distinct) ungapped input row is spelt by exactly one path.  rows: iterable of str over ACGT-."""
    tree = parse_prg(prg)
    _prepare(tree)
    distinct = list(dict.fromkeys(r.replace("-", "") for r in rows))
    if sample is not None:
        distinct = distinct[::max(1, len(distinct) // sample)]
    bad = [(i, n) for i, n in ((i, spellings(tree, r)) for i, r in enumerate(distinct)) if n != 1]
    assert not bad, f"{len(bad)} of {len(distinct)} rows are not spelt exactly once, first (index, paths): {bad[:5]}"
    return len(distinct)

"""PRG string → GFA1 text (API of make_prg/utils/gfa.py).

GFA_Output.gfa_text() is a single pass over the PRG's tokens with an explicit stack (O(length)); the reference's
recursive splitter (build_gfa_string, O(length^2) string concatenation and one substring search per site) is kept as the
definition of the output: the single pass only handles PRGs whose markers are exactly what PrgBuilder emits (sites
5, 7, 9, ... opened in text order, properly nested, every site with two or more alleles, texts free of digits and spaces)
and hands anything else — where the reference's `str(site) in prg_string` substring test or its assertions could fire —
to the recursive form, untouched."""
import re

_MARKER = re.compile(r" (\d+) ")
_DNA_ONLY = re.compile(r"[^0-9 ]*\Z")


class GFA_Output:
    def __init__(self, gfa_string="", gfa_id=0, gfa_site=5):
        self.gfa_string = gfa_string
        self.gfa_id = gfa_id
        self.gfa_site = gfa_site
        self.delim_char = " "

    def split_on_site(self, prg_string, site_num):
        delim = f"{self.delim_char}{site_num}{self.delim_char}"
        parts = prg_string.split(delim)      # non-overlapping leftmost matches of the literal marker
        assert delim.join(parts) == prg_string
        return parts

    def _segment(self, text):
        self.gfa_string += "S\t%d\t%s\tRC:i:0\n" % (self.gfa_id, text if text != "" else "*")

    def _link(self, a, b):
        self.gfa_string += "L\t%d\t+\t%d\t+\t0M\n" % (a, b)

    def build_gfa_string(self, prg_string, pre_var_id=None):
        """Recursive site splitting (reference :39-97); the substring test on the bare site number is kept."""
        end_ids = []
        while str(self.gfa_site) in prg_string:
            prgs = self.split_on_site(prg_string, self.gfa_site)
            assert len(prgs) == 3, "Invalid prg sequence %s for site %d and id %d" % (prg_string, self.gfa_site, self.gfa_id)
            self._segment(prgs[0])
            pre_var_id = self.gfa_id
            self.gfa_id += 1
            for e in end_ids:
                self._link(e, pre_var_id)
                end_ids = []
            alleles = self.split_on_site(prgs[1], self.gfa_site + 1)
            assert len(alleles) > 1, "Invalid prg sequence %s for site %d and id %d" % (prg_string, self.gfa_site + 1, self.gfa_id)
            self.gfa_site += 2
            for allele in alleles:
                if pre_var_id is not None:
                    self._link(pre_var_id, self.gfa_id)
                end_ids.extend(self.build_gfa_string(prg_string=allele, pre_var_id=pre_var_id))
            prg_string = prgs[2]
            pre_var_id = None
        self._segment(prg_string)
        for e in end_ids:
            self._link(e, self.gfa_id)
        ret = [self.gfa_id]
        self.gfa_id += 1
        return ret

    @staticmethod
    def gfa_text(prg_string) -> str:
        from . import native
        fast = native.gfa_text(prg_string)               # libmprg's one-pass host builder (same procedure, C)
        if fast is not None:
            return fast.decode("ascii")
        fast = gfa_text_single_pass(prg_string)
        if fast is not None:
            return fast
        g = GFA_Output(GFA_HEADER)
        g.build_gfa_string(prg_string=prg_string)
        return g.gfa_string

    @staticmethod
    def gfa_bytes(prg_string) -> bytes:
        """The same text as bytes (what the containers store), without a decode/encode round trip."""
        from . import native
        fast = native.gfa_text(prg_string)
        return fast if fast is not None else GFA_Output.gfa_text(prg_string).encode()

    @staticmethod
    def write_gfa(prefix, prg_string):
        with open(f"{prefix}.gfa", "w") as f:
            f.write(GFA_Output.gfa_text(prg_string))


GFA_HEADER = "H\tVN:Z:1.0\tbn:Z:--linear --singlearr\n"


def gfa_text_single_pass(prg: str):
    """The text build_gfa_string() produces, or None if `prg` is not a plain PrgBuilder string (see module docstring).
    Every text between markers becomes one S line, ids in text order; a site's opening text links to the first segment
    of each allele, the last segment of each allele links to the text after the site (reference utils/gfa.py:39-97)."""
    pieces = _MARKER.split(prg)                  # text, marker, text, marker, ..., text (leftmost non-overlapping markers)
    texts, marks = pieces[0::2], pieces[1::2]
    out = [GFA_HEADER]
    add = out.append
    next_site = 5
    sites = []                                   # open sites: [site number, id of the opening text, ends of closed alleles]
    pending = [[]]                               # per open string: allele ends of the site just closed in it
    n = len(marks)
    for i, text in enumerate(texts):
        if _DNA_ONLY.match(text) is None:
            return None
        add("S\t%d\t%s\tRC:i:0\n" % (i, text if text else "*"))
        for e in pending[-1]:
            add("L\t%d\t+\t%d\t+\t0M\n" % (e, i))
        pending[-1] = []
        if i == n:
            break
        m = int(marks[i])
        if m & 1 and sites and m == sites[-1][0]:            # the site closes: this text ended its last allele
            site = sites.pop()
            if not site[2]:
                return None                                   # a single allele: the reference asserts
            pending.pop()
            site[2].append(i)
            pending[-1] = site[2]
        elif m & 1:                                           # a new site opens after this text
            if m != next_site:
                return None
            next_site += 2
            sites.append([m, i, []])
            add("L\t%d\t+\t%d\t+\t0M\n" % (i, i + 1))
            pending.append([])
        else:                                                 # next allele of the innermost open site
            if not sites or m != sites[-1][0] + 1:
                return None
            sites[-1][2].append(i)
            add("L\t%d\t+\t%d\t+\t0M\n" % (sites[-1][1], i + 1))
            pending[-1] = []
    if sites:
        return None
    return "".join(out)

"""PRG string → GFA1 text (API of make_prg/utils/gfa.py)."""


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
        g = GFA_Output("H\tVN:Z:1.0\tbn:Z:--linear --singlearr\n")
        g.build_gfa_string(prg_string=prg_string)
        return g.gfa_string

    @staticmethod
    def write_gfa(prefix, prg_string):
        with open(f"{prefix}.gfa", "w") as f:
            f.write(GFA_Output.gfa_text(prg_string))

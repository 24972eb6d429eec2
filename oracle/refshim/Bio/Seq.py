class Seq(str):
    """str subclass: enough of Bio.Seq.Seq for make_prg (upper, slicing, indexing, iteration)."""

    def __new__(cls, data=""):
        return super().__new__(cls, str(data))

    def upper(self):
        return Seq(str.upper(self))

    def __getitem__(self, item):
        r = str.__getitem__(self, item)
        return Seq(r) if isinstance(item, slice) else r

    def __repr__(self):
        return f"Seq('{str(self)}')"

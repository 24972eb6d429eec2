class UnknownOutputTypeError(Exception):
    pass


class OutputType:
    """-O flag set: p(rg), b(inary), g(fa), a(ll) — reference subcommands/output_type.py:5-33."""
    BINARY, ALL, PRG, GFA = "b", "a", "p", "g"

    def __init__(self, value: str):
        self.type = set(value.lower())
        if not (self.prg or self.binary or self.gfa):
            raise UnknownOutputTypeError(f"{value} is an unknown output type")

    def _all(self) -> bool:
        return self.ALL in self.type

    @property
    def prg(self) -> bool:
        return self._all() or self.PRG in self.type

    @property
    def binary(self) -> bool:
        return self._all() or self.BINARY in self.type

    @property
    def gfa(self) -> bool:
        return self._all() or self.GFA in self.type

"""Multiple-sequence aligners of `make_prg update` (reference make_prg/utils/msa_aligner.py:1-142).

`LeafNode._update_leaf` hands the aligner a leaf's current alignment and the set of new sequences and expects the
alignment with those sequences added (MAFFT --add).  The reference bundles a prebuilt MAFFT; this package does not ship
binaries: `MAFFT` drives an executable named by $MAKE_PRG_MAFFT or found on PATH (`mafft`), with the reference's exact
arguments.  `ReplayAligner` answers from a recorded table instead (previous alignment + new sequences -> updated
alignment): the GPU test box has no MAFFT, and a recorded run of the real aligner is what makes `update` reproducible
there (tests/golden/update.json.gz holds the calls of the reference's own update test cases)."""
import hashlib
import json
import logging
import os
import shutil
import subprocess
import tempfile
import time
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Set

from ..msa import MSA, Record, load_alignment_file

logger = logging.getLogger("make_prg_amd")


class NotAValidExecutableError(Exception):
    pass


class ExecutionError(Exception):
    pass


def new_sequences_fasta(new_sequences: Iterable[str]) -> str:
    """new_sequences.fa of the reference (:75-88): sorted, named Denovo_path_<i>."""
    return "".join(f">Denovo_path_{i}\n{s}\n" for i, s in enumerate(sorted(new_sequences)))


class MSAAligner(ABC):
    def __init__(self, executable: str, tmpdir: Path = Path("..")):
        if shutil.which(executable, mode=os.X_OK) is None:
            raise NotAValidExecutableError(f"Given MSA executable {executable} does not work or is invalid")
        self._executable = executable
        tmpdir.mkdir(parents=True, exist_ok=True)
        self._tmpdir = tmpdir

    @abstractmethod
    def get_updated_alignment(self, current_alignment: MSA, new_sequences: Set[str]) -> MSA:
        raise NotImplementedError

    @classmethod
    def get_aligner_name(cls) -> str:
        raise NotImplementedError

    def _run_aligner(self, args: str, env=None):
        start = time.time()
        process = subprocess.Popen(args, stderr=subprocess.PIPE, encoding="utf-8", shell=True, env=env)
        _, err = process.communicate()
        if process.returncode != 0:
            raise ExecutionError(f"Failed to execute {self.get_aligner_name()} for arguments {args} due to the "
                                 f"following error:\n{err}")
        logger.debug(f"{self.get_aligner_name()} runtime for arguments {args} in seconds: {time.time() - start:.3f}")


class MAFFT(MSAAligner):
    """mafft --auto --quiet --thread 1 --add new_sequences.fa previous_msa.fa > updated_msa.fa (reference :90-142)."""

    def __init__(self, tmpdir: Path = Path(".."), executable: Optional[str] = None):
        super().__init__(executable or os.environ.get("MAKE_PRG_MAFFT") or "mafft", tmpdir)

    @classmethod
    def get_aligner_name(cls) -> str:
        return "MAFFT"

    def get_updated_alignment(self, current_alignment: MSA, new_sequences: Set[str]) -> MSA:
        run_tmpdir = Path(tempfile.mkdtemp(dir=self._tmpdir))
        try:
            previous, added, updated = run_tmpdir / "previous_msa.fa", run_tmpdir / "new_sequences.fa", run_tmpdir / "updated_msa.fa"
            previous.write_text(format(current_alignment, "fasta"))
            added.write_text(new_sequences_fasta(new_sequences))
            args = " ".join([self._executable, "--auto", "--quiet", "--thread", "1", "--add", str(added), str(previous), ">",
                             str(updated)])
            self._run_aligner(args, dict(os.environ, TMPDIR=str(run_tmpdir)))
            return load_alignment_file(str(updated), "fasta")
        finally:
            shutil.rmtree(run_tmpdir, ignore_errors=True)


class ReplayAligner:
    """Answers get_updated_alignment() from recorded aligner calls.  A record is {previous_msa: FASTA text of the current
    alignment as the aligner was given it, new_sequences: sorted list, updated_rows: [[id, description, sequence], ...]
    of the aligner's output after load_alignment_file}.  An unknown request raises KeyError: a replay never guesses."""

    def __init__(self, records: List[dict]):
        self._table: Dict[str, dict] = {self._key(r["previous_msa"], r["new_sequences"]): r for r in records}
        self.calls = 0

    @staticmethod
    def _key(previous_msa: str, new_sequences: Iterable[str]) -> str:
        h = hashlib.sha256(previous_msa.encode())
        h.update(b"\0" + "\n".join(sorted(new_sequences)).encode())
        return h.hexdigest()

    @classmethod
    def from_file(cls, path) -> "ReplayAligner":
        with open(path) as fh:
            return cls(json.load(fh))

    @classmethod
    def get_aligner_name(cls) -> str:
        return "replay"

    def get_updated_alignment(self, current_alignment: MSA, new_sequences: Set[str]) -> MSA:
        key = self._key(format(current_alignment, "fasta"), new_sequences)
        if key not in self._table:
            raise KeyError("ReplayAligner: no recorded aligner call for this alignment and these new sequences "
                           f"({len(current_alignment)} rows + {sorted(new_sequences)})")
        self.calls += 1
        return MSA([Record(seq, rid, desc) for rid, desc, seq in self._table[key]["updated_rows"]])

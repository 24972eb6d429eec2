"""File helpers of the from_msa driver (reference utils/io_utils.py, utils/input_output_files.py:233-235)."""
import re
from pathlib import Path
from typing import Dict
from zipfile import ZipFile, ZipInfo

from ..msa import load_alignment_file  # noqa: F401  (same name as the reference helper)


def remove_known_input_extensions(filename: str) -> str:
    return re.sub(r"\.(fa|fasta)(\.gz)?$", "", filename)


def output_files_already_exist(output_type, output_prefix: str) -> bool:
    names = []
    if output_type.prg:
        names += [".prg.fa", ".update_DS.zip"]
    if output_type.gfa:
        names += [".prg.gfa", ".prg.gfa.zip"]
    if output_type.binary:
        names += [".prg.bin", ".prg.bin.zip"]
    return any(Path(output_prefix + n).exists() for n in names)


def zip_bytes(zip_filepath: Path, name_to_bytes: Dict[str, bytes]):
    assert Path(zip_filepath).suffix == ".zip", "zip_bytes() was not given a .zip filepath"
    with ZipFile(zip_filepath, "w") as z:
        for name, data in name_to_bytes.items():
            z.writestr(ZipInfo(name, date_time=(1980, 1, 1, 0, 0, 0)), data)

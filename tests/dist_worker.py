"""Worker for the world_size-2 test: from_msa.run under torch.distributed (gloo on CPU, emulation backend)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MPRG_DIST_BACKEND", "gloo")

from make_prg_amd import device  # noqa: E402
from make_prg_amd.subcommands import from_msa  # noqa: E402
from make_prg_amd.subcommands.output_type import OutputType  # noqa: E402
from tests.emu.backend import EmuBackend  # noqa: E402

if __name__ == "__main__":
    inp, prefix = sys.argv[1], sys.argv[2]
    device.set_backend(EmuBackend())
    opts = argparse.Namespace(input=inp, suffix="", output_prefix=prefix, alignment_format="fasta", log=None,
                              max_nesting=5, min_match_length=7, output_type=OutputType("a"), force=True, threads=1,
                              verbose=False)
    from_msa.run(opts)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()

"""make_prg-compatible command line for the from_msa path (flags as make_prg/__main__.py:13-96)."""
import argparse
import logging
import os
import sys

from . import __version__
from .subcommands import from_msa, update
from .subcommands.output_type import OutputType


MAX_WORKERS_PER_GPU = 10


def main(argv=None):
    parser = argparse.ArgumentParser(prog="make_prg", usage="make_prg <subcommand> <options>",
                                     description="Subcommand entrypoint")
    parser.add_argument("-V", "--version", action="version", version=__version__)
    subparsers = parser.add_subparsers(title="Available subcommands", help="", metavar="")
    msa_parser = from_msa.register_parser(subparsers)
    update_parser = update.register_parser(subparsers)
    for par in (msa_parser, update_parser):
        par.add_argument("-O", "--output-type", default="a", type=OutputType,
                         help="p: PRG, b: Binary, g: GFA, a: All. Combinations are allowed i.e., gb: GFA and Binary. "
                              "Default: %(default)s")
        par.add_argument("-F", "--force", action="store_true", default=False, help="Force overwrite previous output")
        par.add_argument("-t", "--threads", action="store", type=int, default=1,
                         help="Number of threads (from_msa: host threads of the parser / encoders / writers on one GPU, host worker "
                              "processes per GPU under torchrun; update: concurrent aligner calls). 0 will use all available. "
                              "Default: %(default)d")
        par.add_argument("-v", "--verbose", action="count", default=0, help="Increase output verbosity")
        par.add_argument("--log", help="Path to write log to. Default is stderr")
    args = parser.parse_args(argv)
    if hasattr(args, "func"):
        level = [logging.INFO, logging.DEBUG, logging.DEBUG][min(args.verbose, 2)]
        logging.basicConfig(level=level, **({"filename": args.log} if args.log else {"stream": sys.stderr}))
        if args.threads == 0:
            # "all available": the CPUs this process may use, shared by the ranks of a node.  from_msa: host threads of the
            # native batch stages (one GPU) or host worker processes per GPU (several ranks), where more than ~10 stop paying;
            # update: threads that wait for aligner processes — no such cap
            local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
            from .utils.misc import effective_cpus
            args.threads = max(1, effective_cpus() // local_world)
            if args.func is from_msa.run and local_world > 1:
                args.threads = min(args.threads, MAX_WORKERS_PER_GPU)
        args.func(args)
    else:
        parser.print_help()


if __name__ == "__main__":
    main()
    if os.environ.get("MPRG_FAST_EXIT", "1") != "0" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # every output file is closed by now: skip the interpreter's and the runtimes' tear-down (un-pinning GBs of buffers).
        # Only the single-process run: a rank of a torchrun job leaves through the normal shutdown, after its sub-command has
        # synchronised the device, met the other ranks at a barrier and destroyed the process group.
        logging.shutdown()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)

"""make_prg-compatible command line for the from_msa path (flags as make_prg/__main__.py:13-96)."""
import argparse
import logging
import os
import sys

from . import __version__
from .subcommands import from_msa
from .subcommands.output_type import OutputType


def main(argv=None):
    parser = argparse.ArgumentParser(prog="make_prg", usage="make_prg <subcommand> <options>",
                                     description="Subcommand entrypoint")
    parser.add_argument("-V", "--version", action="version", version=__version__)
    subparsers = parser.add_subparsers(title="Available subcommands", help="", metavar="")
    msa_parser = from_msa.register_parser(subparsers)
    for par in (msa_parser,):
        par.add_argument("-O", "--output-type", default="a", type=OutputType,
                         help="p: PRG, b: Binary, g: GFA, a: All. Combinations are allowed i.e., gb: GFA and Binary. "
                              "Default: %(default)s")
        par.add_argument("-F", "--force", action="store_true", default=False, help="Force overwrite previous output")
        par.add_argument("-t", "--threads", action="store", type=int, default=1,
                         help="Host worker processes per GPU (each builds a part of the rank's alignments on the rank's "
                              "device); 0: one per CPU. Default: %(default)d")
        par.add_argument("-v", "--verbose", action="count", default=0, help="Increase output verbosity")
        par.add_argument("--log", help="Path to write log to. Default is stderr")
    args = parser.parse_args(argv)
    if hasattr(args, "func"):
        level = [logging.INFO, logging.DEBUG, logging.DEBUG][min(args.verbose, 2)]
        logging.basicConfig(level=level, **({"filename": args.log} if args.log else {"stream": sys.stderr}))
        if args.threads == 0:
            args.threads = os.cpu_count()
        args.func(args)
    else:
        parser.print_help()


if __name__ == "__main__":
    main()

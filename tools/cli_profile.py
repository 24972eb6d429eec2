"""GPU-box helper: cProfile of the command line's per-rank work (ingest, build, per-locus outputs) in one process."""
import argparse
import cProfile
import os
import pstats
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathlib import Path
from make_prg_amd.subcommands import from_msa
from make_prg_amd.subcommands.output_type import OutputType
from make_prg_amd.utils.synthetic import synth_config_fasta

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ot = sys.argv[2] if len(sys.argv) > 2 else "a"
with tempfile.TemporaryDirectory() as tmp:
    d = Path(tmp) / "msas"
    d.mkdir()
    for s in range(n):
        (d / f"gene{s}.fa").write_text(synth_config_fasta("C", s))
    opts = argparse.Namespace(input=str(d), suffix="", output_prefix=str(Path(tmp) / "o" / "x"), alignment_format="fasta", log=None,
                              max_nesting=5, min_match_length=7, output_type=OutputType(ot), force=True, threads=1, verbose=False)
    (Path(tmp) / "o").mkdir()
    files = sorted(d.iterdir())
    from_msa.build_shard(files[:20], opts)          # warm-up (library load, first allocations)
    pr = cProfile.Profile()
    pr.enable()
    out = from_msa.build_shard(files, opts)
    from_msa.write_final_files(out, opts.output_type, opts.output_prefix)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

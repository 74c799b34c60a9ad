"""Command line of the MI355X build: `newmap index | search | track` with the reference's flags
(reference: newmap/main.py:30-247).  `python -m newmap_amd.main ...` or the `newmap` console
script."""
from __future__ import annotations

import sys
from argparse import ArgumentParser

from . import __version__, index, search
from .util import DEFAULT_MAPPABILITY_READ_LENGTH, INDEX_EXTENSION

DEFAULT_COMPRESSION_RATIO = 8            # newmap/main.py:13
DEFAULT_SEED_LENGTH = 12                 # newmap/main.py:14
DEFAULT_KMER_BATCH_SIZE = 10000000       # newmap/main.py:18
DEFAULT_THREAD_COUNT = 1                 # newmap/main.py:19
DEFAULT_KMER_SEARCH_RANGE = "20:200"     # newmap/main.py:20
STDOUT_FILENAME = "-"

INDEX_SUBCOMMAND = "index"
UNIQUE_LENGTHS_SUBCOMMAND = "search"
GENERATE_MAPPABILITY_SUBCOMMAND = "track"


def _track_main(args):
    from . import track
    track.main(args)


def build_parser() -> ArgumentParser:
    parser = ArgumentParser(prog="newmap",
                            description="Newmap: A tool for generating mappability data for a reference "
                                        "sequence (MI355X engine)")
    parser.add_argument("--version", action="version", version=f"%(prog)s {__version__} (newmap_amd)")
    sub = parser.add_subparsers(title="subcommands, to be run in order", metavar="", required=True)

    p = sub.add_parser(INDEX_SUBCOMMAND, help="Create an FM index from sequences")
    p.set_defaults(func=index.main)
    p.add_argument("fasta_file", help="Reference sequence file in FASTA format")
    p.add_argument("--output", "-i", metavar="FILE",
                   help=f"Filename of the index file to write. (default: fasta_file with the extension "
                        f"changed to '.{INDEX_EXTENSION}')")
    g = p.add_argument_group("performance tuning arguments")
    g.add_argument("--compression-ratio", "-c", type=int, default=DEFAULT_COMPRESSION_RATIO, metavar="RATIO",
                   help="Suffix array sampling ratio; recorded for interface parity, counting does not "
                        f"use a sampled suffix array. (default: {DEFAULT_COMPRESSION_RATIO})")
    g.add_argument("--seed-length", "-s", type=int, default=DEFAULT_SEED_LENGTH, metavar="LENGTH",
                   help="Length of k-mers memoized in the device seed table (4^LENGTH entries of 8 bytes "
                        f"in HBM, built when the index is opened). (default: {DEFAULT_SEED_LENGTH})")
    g.add_argument("--device", "-d", type=int, default=None, metavar="GPU",
                   help="sort the suffixes on this MI355X instead of the host cores (same index file)")

    p = sub.add_parser(UNIQUE_LENGTHS_SUBCOMMAND,
                       help="Find the shortest unique sequence length at each position in sequences")
    p.set_defaults(func=search.main)
    p.add_argument("fasta_file", metavar="fasta_file", help="(gzipped) fasta file for kmer generation")
    p.add_argument("index_file", nargs="?",
                   help=f"index file to count occurrences in (default: basename of fasta_file with the "
                        f"{INDEX_EXTENSION} extension)")
    g = p.add_argument_group("output arguments")
    g.add_argument("--search-range", "-r", metavar="RANGE", default=DEFAULT_KMER_SEARCH_RANGE,
                   help="Comma separated list of lengths, or an inclusive range separated by a colon. "
                        f"Examples: 20,24,30 or 20:30. (default: {DEFAULT_KMER_SEARCH_RANGE})")
    g.add_argument("--output-directory", "-o", metavar="DIR", default=".",
                   help="Directory for the 'unique' binary files. (default: current working directory)")
    g.add_argument("--include-sequences", "-i", metavar="IDS", help="comma separated sequence IDs to select")
    g.add_argument("--exclude-sequences", "-x", metavar="IDS", help="comma separated sequence IDs to exclude")
    g.add_argument("--norc", action="store_true", help="do not search the reverse-complement strand")
    g.add_argument("--verbose", "-v", action="store_true", help="Print additional information to standard error")
    g = p.add_argument_group("performance arguments")
    g.add_argument("--initial-search-length", "-l", type=int, metavar="LENGTH", default=0,
                   help="accepted for compatibility: it shaped the reference's probe schedule, never the result")
    g.add_argument("--kmer-batch-size", "-s", default=DEFAULT_KMER_BATCH_SIZE, metavar="SIZE", type=int,
                   help=f"Maximum number of positions per device launch. (default: {DEFAULT_KMER_BATCH_SIZE})")
    g.add_argument("--num-threads", "-t", default=DEFAULT_THREAD_COUNT, metavar="NUM", type=int,
                   help="accepted for compatibility (the reference's OpenMP team size)")
    g.add_argument("--device", "-d", type=int, default=None, metavar="GPU",
                   help="MI355X device index (default: LOCAL_RANK or 0)")

    p = sub.add_parser(GENERATE_MAPPABILITY_SUBCOMMAND,
                       help="Calculate single and multi-read mappability tracks from shortest unique lengths")
    p.set_defaults(func=_track_main)
    p.add_argument("read_length", nargs="?", default=str(DEFAULT_MAPPABILITY_READ_LENGTH), metavar="read_length",
                   help=f"read length (default is {DEFAULT_MAPPABILITY_READ_LENGTH})")
    p.add_argument("unique_count_files", nargs="+", help="One or more unique count files")
    g = p.add_argument_group("output arguments")
    g.add_argument("--single-read", "-s", metavar="FILE", help="single-read mappability BED output")
    g.add_argument("--multi-read", "-m", metavar="FILE", help="multi-read mappability WIG output")
    g.add_argument("--verbose", "-v", action="store_true")
    return parser


def parse_subcommands(argv=None):
    parser = build_parser()
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        parser.print_help()
        return
    args = parser.parse_args(argv)
    # a command-line run opens the index once and searches once: small device tables (<= 20 GB, 0.3 - 0.6 s to set up)
    # instead of the throughput-sized ones of a resident handle (up to 180 GB, seconds to allocate).  Override with
    # NEWMAP_AMD_SEED_LENGTH=auto.
    import os
    os.environ.setdefault("NEWMAP_AMD_SEED_LENGTH", "auto-small")
    args.func(args)


if __name__ == "__main__":
    parse_subcommands()

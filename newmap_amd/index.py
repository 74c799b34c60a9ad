"""`newmap index` (reference: newmap/index.py:7-23)."""
from pathlib import Path

from ._c_newmap_generate_index import generate_fm_index
from .util import INDEX_EXTENSION


def main(args):
    index_filename = args.output
    if not index_filename:
        index_filename = Path(args.fasta_file).stem + "." + INDEX_EXTENSION     # index.py:12-15
    generate_fm_index(str(args.fasta_file), str(index_filename), args.compression_ratio, args.seed_length,
                      getattr(args, "device", None))

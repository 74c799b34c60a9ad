"""Small helpers shared by the sub-commands (reference: newmap/util.py)."""
from __future__ import annotations

import gzip
import sys
from pathlib import Path

INDEX_EXTENSION = "awfmi"                   # newmap/util.py:6 (default file name only; the format is ours)
DEFAULT_MAPPABILITY_READ_LENGTH = 24        # newmap/util.py:7


def optional_gzip_open(file_path: Path, mode: str):
    """newmap/util.py:10-18: gzip when the name ends in .gz"""
    file_path = Path(file_path)
    if file_path.suffix == ".gz":
        return gzip.open(file_path, mode)
    return open(file_path, mode)


def verbose_print(verbose: bool, *args):
    if verbose:
        print(*args, file=sys.stderr)

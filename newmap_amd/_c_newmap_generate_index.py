"""Drop-in for the reference's CPython extension ``newmap._c_newmap_generate_index``
(src/newmap-generate-index.c:11-79): same function name, arguments and exceptions."""
from __future__ import annotations

import os

from . import _lib


def generate_fm_index(fasta_filename, index_filename, suffix_array_compression_ratio,
                      kmer_length_in_seed_table, device=None) -> None:
    """generate_fm_index(fasta, index, compression_ratio, seed_length) -> None.

    Raises FileNotFoundError / MemoryError / OSError like src/newmap-generate-index.c:32-54.
    The two integers are taken mod 256 like the reference's "BB" format (:17).  An existing
    index file is overwritten (the behaviour the reference's tests rely on,
    tests/test_unique_counts.py:25-35).  `device` (extension, default None = host builder; or
    NEWMAP_AMD_INDEX_BUILDER=device) computes the suffix array on that GPU; the file is the same."""
    if not isinstance(fasta_filename, (str, os.PathLike)) or not isinstance(index_filename, (str, os.PathLike)):
        raise TypeError("generate_fm_index() arguments 1 and 2 must be str")
    ratio = int(suffix_array_compression_ratio) & 0xFF
    seed = int(kmer_length_in_seed_table) & 0xFF
    if seed > 16:
        raise ValueError(f"seed length {seed} is larger than the supported maximum of 16")
    if device is None and os.environ.get("NEWMAP_AMD_INDEX_BUILDER", "host") == "device":
        device = int(os.environ.get("NEWMAP_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device is None:
        rc = _lib.lib().nm_index_build(os.fsencode(fasta_filename), os.fsencode(index_filename), ratio, seed)
    else:   # engine extension: suffix sort on the GPU (same file)
        rc = _lib.lib().nm_index_build_device(os.fsencode(fasta_filename), os.fsencode(index_filename), ratio, seed,
                                              int(device))
    _lib.raise_for(rc)

"""Drop-in for the reference's CPython extension ``newmap._c_newmap_count_kmers``
(src/newmap-count.c:208-231): same function names, argument meaning, return type (list of int)
and exception classes.  `num_threads` is accepted for interface parity (it sized the OpenMP team
of awFmParallelSearchCount, :79,196); parallelism here is the GPU's."""
from __future__ import annotations

import numpy as np

from .engine import cached_index


def _check_threads(num_threads):
    if not isinstance(num_threads, int):
        raise TypeError("an integer is required (got type %s)" % type(num_threads).__name__)
    if num_threads < 0:
        raise OverflowError("unsigned byte integer is less than minimum")      # format "b", :31-37
    if num_threads > 255:
        raise OverflowError("unsigned byte integer is greater than maximum")


def count_kmers(index_path: str, kmers: list, num_threads: int = 1) -> list:
    """src/newmap-count.c:28-89: forward-strand occurrences of each byte-string k-mer."""
    if not isinstance(index_path, str):
        raise TypeError("argument 1 must be str, not %s" % type(index_path).__name__)
    _check_threads(num_threads)
    if not isinstance(kmers, list):
        raise TypeError("Second argument must be a list of kmer byte strings")          # :40-43
    for k in kmers:
        if not isinstance(k, bytes):
            raise TypeError("All elements of the kmer list must be byte strings")       # :71-76
        if len(k) == 0:
            raise ValueError("All elements of the kmer list must have non-zero length")  # :64-69
    ix = cached_index(index_path)
    return ix.count_kmers(kmers).tolist()


def count_kmers_from_sequence(index_path: str, sequence, starts: list, lengths: list,
                              num_threads: int = 1) -> list:
    """src/newmap-count.c:91-206: forward-strand occurrences of sequence[s:s+l] for each pair."""
    if not isinstance(index_path, str):
        raise TypeError("argument 1 must be str, not %s" % type(index_path).__name__)
    _check_threads(num_threads)
    try:
        seq = memoryview(sequence)
    except TypeError:
        raise TypeError("argument 2 must be read-only bytes-like object, not %s"
                        % type(sequence).__name__) from None
    if not isinstance(starts, list):
        raise TypeError("Third argument must be a list of integers")                    # :111-115
    if not isinstance(lengths, list):
        raise TypeError("Fourth argument must be a list of integers")                   # :117-121
    if len(starts) != len(lengths):
        raise ValueError("Both lists of indices and lengths must be the same length")   # :127-132
    for s, l in zip(starts, lengths):
        if not isinstance(s, int) or not isinstance(l, int):
            raise TypeError("All elements of the the index and length lists must be integers")   # :154-159
        if s < 0:
            raise IndexError("All elements of the the index list must be non-negative integers")  # :163-170
        if l < 0:
            raise ValueError("All lengths in the length list must be non-negative integers")
        if s + l > len(seq):
            raise IndexError("The sum of the index and length of each k-mer must be less than or "
                             "equal to the length of the input byte sequence")           # :184-190
    ix = cached_index(index_path)
    return ix.count_from_sequence(np.frombuffer(seq, dtype=np.uint8), starts, lengths).tolist()

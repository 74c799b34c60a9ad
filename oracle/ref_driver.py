"""oracle/ref_driver.py -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

numpy restatement of the *driver* half of newmap's `search` path, in the reference's own
terms, with the k-mer counts supplied by the C oracle (oracle/kmer_oracle.c).  Every function
cites the reference lines (relative to /root/reference) it follows.

  records / ids ................ newmap/fasta.py:20-106 (header rules :59,75)
  segment geometry ............. newmap/fasta.py:109-190
  num_kmers .................... newmap/search.py:727-741
  ambiguity mask ............... newmap/search.py:744-766
  per-position upper bound ..... newmap/search.py:769-882
  binary search ................ newmap/search.py:383-548
  list / fixed-k search ........ newmap/search.py:551-644
  strand sum + zero guard ...... newmap/search.py:647-724
  per-record output arrays ..... newmap/search.py:197-380

Pinned by tests/test_oracle_golden.py against the reference's known-answer vectors and against
fixtures produced by importing the reference's Python driver (tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes
import gzip
import os
import subprocess
from dataclasses import dataclass
from pathlib import Path
from typing import Iterable, Iterator, Sequence

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liboracle.so"
_lib = None


def build_library(force: bool = False) -> Path:
    """Compile oracle/kmer_oracle.c into oracle/_build/liboracle.so (gcc, OpenMP)."""
    src = _HERE / "kmer_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s", "all"], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        build_library()
    L = ctypes.CDLL(str(_LIB_PATH))
    vp, i64, u32, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int
    L.or_build.restype = vp
    L.or_build.argtypes = [vp, vp, i64]
    L.or_build_ptrs.restype = vp
    L.or_build_ptrs.argtypes = [vp, vp, i64]
    L.or_build_ptrs_depth.restype = vp
    L.or_build_ptrs_depth.argtypes = [vp, vp, i64, i64]
    L.or_drop_sa.argtypes = [vp]
    L.or_free.argtypes = [vp]
    L.or_text_length.restype = i64
    L.or_text_length.argtypes = [vp]
    L.or_suffix_array.restype = vp
    L.or_suffix_array.argtypes = [vp]
    L.or_mapped_text.restype = vp
    L.or_mapped_text.argtypes = [vp]
    L.or_count.restype = u32
    L.or_count.argtypes = [vp, ctypes.c_char_p, i64]
    L.or_fm_count.restype = u32
    L.or_fm_count.argtypes = [vp, ctypes.c_char_p, i64]
    for name in ("or_count_from_sequence", "or_fm_count_from_sequence"):
        f = getattr(L, name)
        f.restype = None
        f.argtypes = [vp, vp, i64, vp, vp, i64, vp]
    L.or_fm_build.restype = i32
    L.or_fm_build.argtypes = [vp, i32]
    L.or_ref_binary_search_segment.restype = i32
    L.or_ref_binary_search_segment.argtypes = [vp, vp, i64, i64, u32, u32, u32, i32, i32,
                                               vp, vp, vp, vp, vp]
    L.or_scan_counts.restype = i32
    L.or_scan_counts.argtypes = [vp, vp, i64, vp, vp, i64, i32, vp]
    L.or_num_threads.restype = i32
    L.or_set_num_threads.argtypes = [i32]
    _lib = L
    return L


# ----------------------------------------------------------------------------- FASTA

HEADER_PREFIXES = (b">", b";")          # newmap/fasta.py:4


def read_records(path_or_lines) -> list[tuple[bytes, bytes]]:
    """Whole-record view of what newmap/fasta.py:20-106 streams: a list of (id, data).

    A line starting with '>' or ';' opens a record whose id is the first whitespace token
    minus its first byte (:75); every other line is right-stripped (:47) and appended; data in
    front of any header belongs to id b'' ; a header without data yields no record (:173-188).
    """
    if isinstance(path_or_lines, (str, os.PathLike)):
        p = Path(path_or_lines)
        opener = gzip.open if p.suffix == ".gz" else open      # newmap/util.py:10-18
        with opener(p, "rb") as fh:
            lines = fh.readlines()
    else:
        lines = list(path_or_lines)
    out: list[tuple[bytes, bytes]] = []
    cur_id, chunks = b"", []

    def flush():
        data = b"".join(chunks)
        if data:
            out.append((cur_id, data))

    for raw in lines:
        line = raw.rstrip()
        if line.startswith(HEADER_PREFIXES):
            flush()
            cur_id, chunks = line.split()[0][1:], []
        else:
            chunks.append(line)
    flush()
    return out


@dataclass
class Segment:                            # newmap/fasta.py:7-17
    id: bytes
    data: bytes
    epilogue: bool
    offset: int = 0                       # position of data[0] inside its record (oracle extra)


def record_segments(rec_id: bytes, data: bytes, length: int, overlap: int = 0) -> list[Segment]:
    """Closed form of newmap/fasta.py:109-190 for one record.

    Segment j covers bytes [j*(length-overlap), j*(length-overlap)+length) while that fits; the
    remainder (overlap included) forms the last segment; the last segment carries `epilogue`.
    If the record ends exactly at a segment end, that full segment is the epilogue (:137-150 of
    tests/test_sequence_buffer_iter.py)."""
    n = len(data)
    if n == 0:
        return []
    step = length - overlap
    segs: list[Segment] = []
    start = 0
    while start + length <= n:
        segs.append(Segment(rec_id, data[start:start + length], False, start))
        start += step
    consumed = segs[-1].offset + length if segs else 0
    if consumed < n:
        segs.append(Segment(rec_id, data[start:], False, start))
    segs[-1].epilogue = True
    return segs


def sequence_segments(path_or_lines, length: int, overlap: int = 0) -> Iterator[Segment]:
    for rec_id, data in read_records(path_or_lines):
        yield from record_segments(rec_id, data, length, overlap)


# ----------------------------------------------------------------------------- index

class OracleIndex:
    """Forward-strand text of all records + suffix array (+ optional FM port)."""

    def __init__(self, records: Sequence, max_depth: int = 0):
        """records: bytes objects or uint8 numpy arrays (not copied: a 3 Gbp genome stays where it is).  The mapped
        text -- one separator per record included -- must stay below 2^32 symbols.  max_depth > 0: order the suffixes
        by their first max_depth symbols only (texts with very long exact repeats): counts of k-mers up to that length
        stay exact, the FM port is not available."""
        self._L = lib()
        self.records = [r for r in records if len(r)]
        views = [np.frombuffer(r, dtype=np.uint8) if isinstance(r, (bytes, bytearray, memoryview)) else np.ascontiguousarray(r, dtype=np.uint8)
                 for r in self.records]
        ptrs = np.array([v.ctypes.data for v in views], dtype=np.uint64)
        lens = np.array([v.size for v in views], dtype=np.int64)
        self._h = self._L.or_build_ptrs_depth(ptrs.ctypes.data if len(views) else None, lens.ctypes.data if len(views) else None, len(views), int(max_depth))
        if not self._h:
            raise MemoryError("oracle index build failed (text of 2^32 symbols or more, or out of memory)")
        self._fm_seed = None

    @classmethod
    def from_fasta(cls, path) -> "OracleIndex":
        return cls([d for _, d in read_records(path)])

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.or_free(h)

    def enable_fm(self, seed_len: int = 12):
        if self._fm_seed != seed_len:
            if self._L.or_fm_build(self._h, seed_len) != 0:
                raise MemoryError("oracle FM port build failed")
            self._fm_seed = seed_len

    def drop_suffix_array(self):
        """after enable_fm(): keep the FM port only (4 bytes per symbol less); or_count then answers 0"""
        self._L.or_drop_sa(self._h)

    # counts at the FFI seam ---------------------------------------------------------
    def count(self, kmer: bytes, fm: bool = False) -> int:
        f = self._L.or_fm_count if fm else self._L.or_count
        return int(f(self._h, bytes(kmer), len(kmer)))

    def count_kmers(self, kmers: Iterable[bytes], fm: bool = False) -> list[int]:
        """src/newmap-count.c:28-89"""
        return [self.count(k, fm) for k in kmers]

    def count_from_sequence(self, seq: bytes, starts, lens, fm: bool = False) -> np.ndarray:
        """src/newmap-count.c:91-206: forward-strand count of seq[s:s+l] for each (s, l)."""
        s = np.ascontiguousarray(starts, dtype=np.int64)
        l = np.ascontiguousarray(lens, dtype=np.int64)
        if s.size and (int((s + l).max()) > len(seq) or int(s.min()) < 0):
            raise IndexError("k-mer outside of the sequence")      # :163-190
        out = np.zeros(s.size, dtype=np.uint32)
        buf = np.frombuffer(bytes(seq), dtype=np.uint8)
        f = self._L.or_fm_count_from_sequence if fm else self._L.or_count_from_sequence
        if s.size:
            f(self._h, buf.ctypes.data, buf.size, s.ctypes.data, l.ctypes.data, s.size,
              out.ctypes.data)
        return out

    @property
    def handle(self):
        return self._h


_COMPLEMENT = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")     # newmap/search.py:22


def scan_total_counts(records: Sequence, seq: bytes, starts, lens, seed: int, use_rc: bool = True) -> np.ndarray:
    """newmap/search.py:647-697 totals (forward + reverse complement) of seq[s:s+l] over `records`, by ONE scan of the
    records with no index (oracle/kmer_oracle.c or_scan_counts): for full-size genomes no suffix array fits.  Every
    length must be >= seed (<= 32)."""
    L = lib()
    starts = np.asarray(starts, dtype=np.int64)
    lens = np.asarray(lens, dtype=np.int64)
    assert lens.size == 0 or int(lens.min()) >= seed
    views = [np.frombuffer(r, dtype=np.uint8) if isinstance(r, (bytes, bytearray, memoryview)) else np.ascontiguousarray(r, dtype=np.uint8)
             for r in records]
    ptrs = np.array([v.ctypes.data for v in views], dtype=np.uint64)
    rlens = np.array([v.size for v in views], dtype=np.int64)
    nq = int(starts.size)
    parts = [seq[int(s):int(s) + int(l)] for s, l in zip(starts, lens)]
    if use_rc:
        parts += [p.translate(_COMPLEMENT)[::-1] for p in parts]
    blob = np.frombuffer(b"".join(parts) or b"\0", dtype=np.uint8)
    off = np.zeros(len(parts) + 1, dtype=np.int64)
    np.cumsum([len(p) for p in parts], out=off[1:])
    counts = np.zeros(max(len(parts), 1), dtype=np.uint32)
    rc = L.or_scan_counts(ptrs.ctypes.data, rlens.ctypes.data, len(views), blob.ctypes.data, off.ctypes.data, len(parts), int(seed),
                          counts.ctypes.data)
    if rc != 0:
        raise MemoryError(f"or_scan_counts failed ({rc})")
    return (counts[:nq] + counts[nq:2 * nq]) if use_rc else counts[:nq]


def closed_form_claims(seq: bytes, out: np.ndarray, kmin: int, kmax: int):
    """What an output array asserts about counts, as (start, length, relation) queries over
    the positions [0, len(out)) of `seq` (which carries its lookahead): out[p] = a > 0 claims total(p, a) == 1 and, for
    a > kmin, total(p, a - 1) >= 2; out[p] == 0 at an unambiguous position with U_p >= kmin claims total(p, U_p) >= 2
    (U_p = min(kmax, unambiguous bases from p on, newmap/search.py:769-882).  Because the total is >= 1 and does not
    grow with the length, these claims hold if and only if `out` is the closed form."""
    n = len(out)
    amb = ~_ALLOWED[np.frombuffer(seq, dtype=np.uint8)]
    idx = np.arange(len(seq), dtype=np.int64)
    nxt = np.where(amb, idx, np.int64(len(seq)))
    nxt = np.minimum.accumulate(nxt[::-1])[::-1]
    room = np.minimum(nxt - idx, kmax)[:n]
    o = out.astype(np.int64)
    uniq = np.flatnonzero(o > 0)
    shorter = uniq[o[uniq] > kmin]
    zero = np.flatnonzero((o == 0) & ~amb[:n] & (room >= kmin))
    assert (o[uniq] <= room[uniq]).all() and (o[uniq] >= kmin).all(), "a reported length lies outside [kmin, U_p]"
    starts = np.concatenate([uniq, shorter, zero])
    lens = np.concatenate([o[uniq], o[shorter] - 1, room[zero]])
    rel = np.concatenate([np.zeros(uniq.size, np.int8), np.ones(shorter.size, np.int8), np.ones(zero.size, np.int8)])   # 0: == 1, 1: >= 2
    return starts, lens, rel


def total_counts(index: OracleIndex, seq: bytes, starts: np.ndarray, lens: np.ndarray,
                 use_rc: bool = True, fm: bool = False) -> np.ndarray:
    """newmap/search.py:647-724 for one index and one sequence."""
    c = index.count_from_sequence(seq, starts, lens, fm).astype(np.uint32)
    if use_rc:
        rc = seq.translate(_COMPLEMENT)[::-1]                       # :682-683
        rstarts = len(seq) - np.asarray(starts, np.int64) - np.asarray(lens, np.int64)   # :687
        c = c + index.count_from_sequence(rc, rstarts, lens, fm)
    if c.size and not np.all(c):                                   # :702-722
        i = int(np.flatnonzero(c == 0)[0])
        s, l = int(starts[i]), int(lens[i])
        raise RuntimeError("The following generated k-mer was not found in the index:\n"
                           f"{seq[s:s + l].decode('utf-8', 'replace')}\n"
                           "Possibly a mismatch between the sequence and the index.")
    return c


# ----------------------------------------------------------------------------- per segment

_ALLOWED = np.zeros(256, dtype=bool)
_ALLOWED[list(b"ACGTacgt")] = True                              # newmap/search.py:23


def num_kmers_of(seg: Segment, kmax: int) -> int:
    """newmap/search.py:727-741"""
    return len(seg.data) if seg.epilogue else len(seg.data) - (kmax - 1)


def ambiguity_mask(data: bytes, num_positions: int) -> np.ndarray:
    """newmap/search.py:744-766: only the first num_positions bytes are looked at."""
    return ~_ALLOWED[np.frombuffer(data, dtype=np.uint8, count=num_positions)]


def upper_search_bound(mask: np.ndarray, kmax: int, buffer_len: int) -> np.ndarray:
    """newmap/search.py:769-882 as a closed form.

    For an unmasked position p: the distance to the next masked position to its right, or to
    the end of the buffer if none follows inside the mask, capped at kmax.  Masked positions keep
    kmax.  Raises AssertionError like :780-784 when the lookahead is too long."""
    n = mask.size
    assert buffer_len - n < kmax, \
        "Excess sequence buffer length is greater than the maximum search length"
    idx = np.arange(n, dtype=np.int64)
    nxt = np.where(mask, idx, np.int64(buffer_len))
    if n:
        nxt = np.minimum.accumulate(nxt[::-1])[::-1]              # nearest masked at/after p
    room = nxt - idx
    return np.where(mask, kmax, np.minimum(room, kmax)).astype(np.int64)


def binary_search_segment(index: OracleIndex, seg: Segment, kmin: int, kmax: int, dtype,
                          use_rc: bool = True, initial_search_length: int = 0,
                          fm: bool = False, log: dict | None = None):
    """newmap/search.py:383-548 for one FASTA / one index (num_sequences == 1)."""
    data = seg.data
    n = num_kmers_of(seg, kmax)
    finished = ambiguity_mask(data, n)
    n_amb = int(finished.sum())
    lower = np.full(n, kmin, dtype=np.int64)
    upper = upper_search_bound(finished, kmax, len(data))
    query = (upper + lower) // 2                                   # :424-426 (exact in float64)
    if initial_search_length:
        query = np.minimum(query, initial_search_length)            # :429-433
    finished = finished | (upper < kmin)                            # :437
    unique = np.zeros(n, dtype=np.int64)
    iters = probes = probe_len = 0
    while not finished.all():                                       # :464
        act = np.flatnonzero(~finished)                             # :472
        q = query[act]
        cnt = total_counts(index, data, act, q, use_rc, fm)         # :475-480
        iters += 1
        probes += act.size
        probe_len += int(q.sum())
        one = cnt == 1
        many = cnt > 1
        u = unique[act]
        unique[act] = np.where(one & ((u == 0) | (q < u)), q, u)    # :489-499
        fin = finished[act]
        fin = np.where(one, q == lower[act], fin)                   # :504-508
        fin = np.where(many, q == upper[act], fin)                  # :513-517
        finished[act] = fin
        upper[act] = np.where(one, q - 1, upper[act])               # :524-527
        lower[act] = np.where(many, q + 1, lower[act])              # :532-535
        query[act] = (upper[act] + lower[act]) // 2                 # :540-542
    if log is not None:
        log["iterations"] = log.get("iterations", 0) + iters
        log["probes"] = log.get("probes", 0) + probes
        log["probe_len"] = log.get("probe_len", 0) + probe_len
    return unique.astype(dtype), n_amb


def linear_search_segment(index: OracleIndex, seg: Segment, kmer_lengths: Sequence[int], kmax: int,
                          dtype, use_rc: bool = True, fm: bool = False):
    """newmap/search.py:551-644 for one FASTA / one index."""
    data = seg.data
    n = num_kmers_of(seg, kmax)
    finished = ambiguity_mask(data, n)
    n_amb = int(finished.sum())
    unique = np.zeros(n, dtype=np.uint32)
    is_N = np.frombuffer(data, dtype=np.uint8) == ord("N")
    n_before = np.concatenate(([0], np.cumsum(is_N)))               # N's in data[:i]
    for k in kmer_lengths:                                          # :578
        act = np.flatnonzero(~finished)
        end = np.minimum(act + k, len(data))                        # slice truncation :590
        has_N = (n_before[end] - n_before[act]) > 0                 # :593 upper-case N only
        finished[act[has_N]] = True                                 # :596
        act = act[~has_N]
        lens = end[~has_N] - act                                    # :600 len(kmer)
        if act.size == 0:                                           # :605-609
            break
        cnt = total_counts(index, data, act, lens, use_rc, fm)      # :615-619
        u = unique[act]
        unique[act] = np.where((cnt == 1) & (u == 0), k, u)         # :627-636
        finished[act] = cnt == 1                                    # :639
    return unique.astype(dtype), n_amb


# ----------------------------------------------------------------------------- per record

def output_dtype(kmax: int):
    """newmap/search.py:204-212"""
    if kmax <= 255:
        return np.uint8, "uint8"
    if kmax <= 65535:
        return np.uint16, "uint16"
    return np.uint32, "uint32"


def unique_counts(fasta, index: OracleIndex, kmer_lengths: Sequence[int], is_binary: bool,
                  batch: int = 10_000_000, use_rc: bool = True, initial_search_length: int = 0,
                  fm: bool = False, log: dict | None = None) -> dict[bytes, np.ndarray]:
    """newmap/search.py:197-380 without the file writes: {record id: unique-length array}.
    Duplicate ids overwrite (:304-305 truncates on a new id)."""
    kmax, kmin = max(kmer_lengths), min(kmer_lengths)
    dtype, _ = output_dtype(kmax)
    lookahead = kmax - 1                                            # :229
    out: dict[bytes, list[np.ndarray]] = {}
    cur = None
    for seg in sequence_segments(fasta, batch + lookahead, lookahead):   # :235,251-255
        if seg.id != cur:
            cur = seg.id
            out[cur] = []
        if is_binary:
            arr, _ = binary_search_segment(index, seg, kmin, kmax, dtype, use_rc,
                                           initial_search_length, fm, log)
        else:
            arr, _ = linear_search_segment(index, seg, kmer_lengths, kmax, dtype, use_rc, fm)
        out[cur].append(arr)
    return {k: (np.concatenate(v) if v else np.zeros(0, dtype)) for k, v in out.items()}


def closed_form_min_unique(record: bytes, index: OracleIndex, kmin: int, kmax: int,
                           use_rc: bool = True, fm: bool = False) -> np.ndarray:
    """the batch-independent statement of the binary mode, i.e. the
    reference driver run with a batch that holds the whole record (one epilogue segment)."""
    dtype, _ = output_dtype(kmax)
    seg = Segment(b"", record, True, 0)
    arr, _ = binary_search_segment(index, seg, kmin, kmax, dtype, use_rc, 0, fm)
    return arr


def ref_binary_search_segment_c(index: OracleIndex, data: bytes, num_kmers: int, kmin: int,
                                kmax: int, use_rc: bool = True, initial_search_length: int = 0,
                                fm: bool = True):
    """The C/OpenMP port of the same schedule (oracle/kmer_oracle.c
    or_ref_binary_search_segment) -- the code bench.py times as `cpu_baseline`."""
    L = lib()
    if fm and index._fm_seed is None:
        raise ValueError("call index.enable_fm(seed_len) before timing the FM port")
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.zeros(max(num_kmers, 1), dtype=np.uint32)
    amb = ctypes.c_int64(0)
    bad_pos, bad_len = ctypes.c_int64(-1), ctypes.c_int64(0)
    stats = np.zeros(3, dtype=np.int64)
    rc = L.or_ref_binary_search_segment(index.handle, buf.ctypes.data, buf.size, num_kmers,
                                        kmin, kmax, initial_search_length, int(use_rc), int(fm),
                                        out.ctypes.data, ctypes.byref(amb), ctypes.byref(bad_pos),
                                        ctypes.byref(bad_len), stats.ctypes.data)
    if rc < 0:
        raise MemoryError("oracle segment search: allocation failed")
    if rc > 0:
        s, l = bad_pos.value, bad_len.value
        raise RuntimeError("The following generated k-mer was not found in the index:\n"
                           f"{data[s:s + l].decode('utf-8', 'replace')}\n"
                           "Possibly a mismatch between the sequence and the index.")
    return out[:num_kmers], int(amb.value), {"probes": int(stats[0]), "probe_len": int(stats[1]),
                                              "iterations": int(stats[2])}


# ----------------------------------------------------------------------------- several FASTA / indexes
# newmap/search.py:251-265 (lock-step segments of several FASTA files), :461 (num_sequences),
# :656-697 (counts summed over index files x sequences).  Mask, upper bound and ids come from the
# FIRST sequence (:263-265, :393-400).

def total_counts_multi(indexes: Sequence[OracleIndex], seqs: Sequence[bytes], starts, lens,
                       use_rc: bool = True) -> np.ndarray:
    c = np.zeros(len(starts), dtype=np.uint32)
    for ix in indexes:                                              # :656
        for seq in seqs:                                            # :659
            c = c + ix.count_from_sequence(seq, starts, lens)
            if use_rc:
                rc = seq.translate(_COMPLEMENT)[::-1]
                c = c + ix.count_from_sequence(rc, len(seq) - np.asarray(starts, np.int64) - np.asarray(lens, np.int64), lens)
    if c.size and not np.all(c):                                   # :702-722
        raise RuntimeError("The following generated k-mer was not found in the index")
    return c


def binary_search_segments_multi(indexes, segs: Sequence[Segment], kmin: int, kmax: int, dtype,
                                 use_rc: bool = True, max_iterations: int = 64):
    first = segs[0]
    ns = len(segs)                                                  # :461
    n = num_kmers_of(first, kmax)
    finished = ambiguity_mask(first.data, n)
    n_amb = int(finished.sum())
    lower = np.full(n, kmin, dtype=np.int64)
    upper = upper_search_bound(finished, kmax, len(first.data))
    query = (upper + lower) // 2
    finished = finished | (upper < kmin)
    unique = np.zeros(n, dtype=np.int64)
    it = 0
    while not finished.all():
        it += 1
        if it > max_iterations:
            raise RuntimeError("a total strictly between 0 and num_sequences: the reference never terminates (SURVEY A.3.7)")
        act = np.flatnonzero(~finished)
        q = query[act]
        cnt = total_counts_multi(indexes, [s.data for s in segs], act, q, use_rc)
        one = cnt == ns                                             # :489-490
        many = cnt > ns                                             # :513-514
        u = unique[act]
        unique[act] = np.where(one & ((u == 0) | (q < u)), q, u)
        fin = finished[act]
        fin = np.where(one, q == lower[act], fin)
        fin = np.where(many, q == upper[act], fin)
        finished[act] = fin
        upper[act] = np.where(one, q - 1, upper[act])
        lower[act] = np.where(many, q + 1, lower[act])
        query[act] = (upper[act] + lower[act]) // 2
    return unique.astype(dtype), n_amb


def linear_search_segments_multi(indexes, segs: Sequence[Segment], kmer_lengths: Sequence[int], kmax: int,
                                 dtype, use_rc: bool = True):
    first = segs[0]
    ns = len(segs)
    data = first.data
    n = num_kmers_of(first, kmax)
    finished = ambiguity_mask(data, n)
    n_amb = int(finished.sum())
    unique = np.zeros(n, dtype=np.uint32)
    is_N = np.frombuffer(data, dtype=np.uint8) == ord("N")
    n_before = np.concatenate(([0], np.cumsum(is_N)))
    for k in kmer_lengths:
        act = np.flatnonzero(~finished)
        end = np.minimum(act + k, len(data))
        has_N = (n_before[end] - n_before[act]) > 0
        finished[act[has_N]] = True
        act = act[~has_N]
        lens = end[~has_N] - act
        if act.size == 0:
            break
        cnt = total_counts_multi(indexes, [s.data for s in segs], act, lens, use_rc)
        u = unique[act]
        unique[act] = np.where((cnt == ns) & (u == 0), k, u)         # :627-636
        finished[act] = cnt == 1                                     # :639 -- literal 1
    return unique.astype(dtype), n_amb


def unique_counts_multi(fastas: Sequence, indexes: Sequence[OracleIndex], kmer_lengths: Sequence[int],
                        is_binary: bool, batch: int = 10_000_000, use_rc: bool = True) -> dict[bytes, np.ndarray]:
    kmax, kmin = max(kmer_lengths), min(kmer_lengths)
    dtype, _ = output_dtype(kmax)
    lookahead = kmax - 1
    streams = [list(sequence_segments(f, batch + lookahead, lookahead)) for f in fastas]
    out: dict[bytes, list[np.ndarray]] = {}
    cur = None
    for segs in zip(*streams):                                       # :260
        if segs[0].id != cur:
            cur = segs[0].id
            out[cur] = []
        if is_binary:
            arr, _ = binary_search_segments_multi(indexes, segs, kmin, kmax, dtype, use_rc)
        else:
            arr, _ = linear_search_segments_multi(indexes, segs, kmer_lengths, kmax, dtype, use_rc)
        out[cur].append(arr)
    return {k: np.concatenate(v) for k, v in out.items()}

"""Test-only host simulation of newmap_amd/csrc/nm_core.h (see hostsim.cpp header)."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libhostsim.so"
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    src = _HERE / "hostsim.cpp"
    core = _HERE.parent.parent / "newmap_amd" / "csrc" / "nm_core.h"
    scan = core.parent / "nm_fasta_scan.hpp"
    stale = (not _SO.exists() or _SO.stat().st_mtime < max(src.stat().st_mtime, core.stat().st_mtime, scan.stat().st_mtime))
    if stale:
        _SO.parent.mkdir(exist_ok=True)
        subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-Wno-unknown-pragmas", "-o", str(_SO), str(src)],
                       check=True)
    L = ctypes.CDLL(str(_SO))
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    L.hs_open.restype = vp
    L.hs_open.argtypes = [ctypes.c_char_p, i32, i32]
    L.hs_close.argtypes = [vp]
    L.hs_enable_lfb.argtypes = [vp, i32]
    L.hs_enable_lf2.argtypes = [vp, i32]
    L.hs_check_lf2.restype = ctypes.c_uint64
    L.hs_check_lf2.argtypes = [vp]
    L.hs_check_bi.restype = u64
    L.hs_check_bi.argtypes = [vp, vp, u64, u64, u64]
    L.hs_set_sweep.argtypes = [i32]
    L.hs_enable_lcp.restype = i32
    L.hs_enable_lcp.argtypes = [vp, i32]
    L.hs_lcp_byte.restype = ctypes.c_int64
    L.hs_lcp_byte.argtypes = [vp, u64]
    L.hs_check_lfb.restype = u64
    L.hs_check_lfb.argtypes = [vp]
    L.hs_check_levels.restype = u64
    L.hs_check_levels.argtypes = [vp, u32]
    L.hs_check_codes4.restype = u64
    L.hs_check_codes4.argtypes = [u64]
    L.hs_check_quad.restype = u64
    L.hs_check_quad.argtypes = [vp]
    L.hs_info.restype = u64
    L.hs_info.argtypes = [vp, i32]
    L.hs_min_unique.restype = i32
    L.hs_min_unique.argtypes = [vp, vp, u64, u64, u32, u32, i32, i32, vp, vp]
    L.hs_repeat_probes.restype = u64
    L.hs_repeat_probes.argtypes = [vp, vp, u64, u64, u32, u32, u32, u32, vp, vp]
    L.hs_repeat_probes2.restype = u64
    L.hs_repeat_probes2.argtypes = [vp, vp, u64, u64, u32, u32, u32, u32, vp, vp, i32, vp, vp]
    L.hs_segment_hash.restype = u64
    L.hs_segment_hash.argtypes = [vp, u64, u64]
    L.hs_guard.restype = u64
    L.hs_guard.argtypes = [vp, vp, u64, u64, vp, u32, i32, u32, i32]
    L.hs_build_dict.restype = ctypes.c_int64
    L.hs_build_dict.argtypes = [vp, u32, u32]
    L.hs_dict_lookup.restype = ctypes.c_int64
    L.hs_dict_lookup.argtypes = [vp, ctypes.c_char_p]
    L.hs_fasta_open.restype = vp
    L.hs_fasta_open.argtypes = [ctypes.c_char_p, u32, u64]
    L.hs_fasta_close.argtypes = [vp]
    L.hs_fasta_count.restype = u64
    L.hs_fasta_count.argtypes = [vp]
    L.hs_fasta_record.restype = u64
    L.hs_fasta_record.argtypes = [vp, u64, ctypes.c_char_p, u64, vp]
    L.hs_fasta_read.argtypes = [vp, u64, u64, u64, vp]
    L.hs_build_quad2.restype = i32
    L.hs_build_quad2.argtypes = [vp, u32]
    L.hs_valid_bits.argtypes = [vp, u64, u32, vp]
    L.hs_sites.restype = i32
    L.hs_sites.argtypes = [vp, vp, u64, u64, u32, u32, u32, i32, vp, u32, i32, vp, vp, vp, vp, u32, u32]
    L.hs_fixed_k.restype = i32
    L.hs_fixed_k.argtypes = [vp, vp, u64, u64, vp, u32, i32, i32, vp, vp]
    L.hs_count.argtypes = [vp, vp, vp, vp, u64, vp]
    L.hs_upper.argtypes = [vp, u64, u64, u32, vp]
    L.hs_multi.restype = i32
    L.hs_multi.argtypes = [vp, u32, vp, u32, u64, u64, u32, u32, vp, u32, i32, i32, vp, vp]
    _lib = L
    return L


def segment_hash(seq: bytes, end: int) -> int:
    """fingerprint of positions [0, end) of a segment, computed from its encoded words as the device does (nm_hash.h)"""
    buf = np.frombuffer(seq, dtype=np.uint8)
    return int(lib().hs_segment_hash(buf.ctypes.data, buf.size, end))


class HostSim:
    def __init__(self, index_path, seed_len=-1, force_big=False):
        self.L = lib()
        self.h = self.L.hs_open(str(index_path).encode(), seed_len, int(force_big))
        if not self.h:
            raise OSError(f"hostsim could not read {index_path}")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hs_close(self.h)
            self.h = None

    def enable_lfb(self, on=True):
        self.L.hs_enable_lfb(self.h, int(on))

    def enable_lf2(self, on=True):
        self.L.hs_enable_lf2(self.h, int(on))

    def check_lf2(self):
        return int(self.L.hs_check_lf2(self.h))

    def check_bi(self, seq: bytes = b"", rounds=2000, seed=12345):
        """nm_bi_extend (rows of a string and of its reverse complement, extended to either side) against plain backward
        searches, on strings that follow `seq` and on random ones; returns the number of disagreements"""
        buf = np.frombuffer(seq, dtype=np.uint8)
        return int(self.L.hs_check_bi(self.h, buf.ctypes.data if buf.size else None, buf.size, rounds, seed))

    def set_sweep(self, on=True):
        """sites() finishes the open positions with the sweep (k_sweep) instead of one walk per position (k_resolve)"""
        self.L.hs_set_sweep(int(on))

    def enable_lcp(self, on=True) -> bool:
        """the sweep reads the index file's LCP bytes where the end of a chain moves (False: it walks); returns whether the file has them"""
        return bool(self.L.hs_enable_lcp(self.h, int(on)))

    def lcp_byte(self, row: int) -> int:
        return int(self.L.hs_lcp_byte(self.h, row))

    def check_lfb(self):
        return int(self.L.hs_check_lfb(self.h))

    def check_levels(self, s):
        return int(self.L.hs_check_levels(self.h, s))

    def check_quad(self):
        """build the quad table from the simulated seed table (core length = seed length <= 8) and compare every
        bit four neighbouring positions can read with a direct count; returns the number of wrong bits"""
        return int(self.L.hs_check_quad(self.h))

    def build_quad2(self, m2):
        """a second quad table with cores of m2 bases: k_resolve asks it before it walks (nm_second_chance)"""
        assert self.L.hs_build_quad2(self.h, m2) == 0

    def info(self, what):
        return int(self.L.hs_info(self.h, what))

    def min_unique(self, seq: bytes, num_kmers, kmin, kmax, use_rc=True, dtype=np.uint8):
        buf = np.frombuffer(seq, dtype=np.uint8)
        out = np.zeros(max(num_kmers, 1), dtype=dtype)
        status = np.zeros(8, dtype=np.uint64)
        rc = self.L.hs_min_unique(self.h, buf.ctypes.data, buf.size, num_kmers, kmin, kmax, int(use_rc),
                                  out.dtype.itemsize, out.ctypes.data, status.ctypes.data)
        return out[:num_kmers], status, rc

    def repeat_probes(self, seq: bytes, num_kmers, kmin, kmax, stride=64, coarse_stride=0):
        """probe word of every stride (newmap_amd/csrc/nm_core.h: nm_repeat_probe), the element the probes decide
        for every position (0xFFFFFFFF = left open), and the LF steps spent."""
        buf = np.frombuffer(seq, dtype=np.uint8)
        n_probes = (num_kmers + stride - 1) // stride
        words = np.zeros(n_probes + 1, dtype=np.uint32)
        decided = np.zeros(max(num_kmers, 1), dtype=np.uint32)
        steps = self.L.hs_repeat_probes(self.h, buf.ctypes.data, buf.size, num_kmers, kmin, kmax, stride, coarse_stride,
                                        words.ctypes.data, decided.ctypes.data)
        return words[:n_probes], decided[:num_kmers], int(steps)

    def repeat_probes_periodic(self, seq: bytes, num_kmers, kmin, kmax, stride=64, coarse_stride=512):
        """the same with the coarse words taken from the tandem runs (k_period_runs / k_period_spread): also returns the period
        found per coarse stride and the coarse words"""
        buf = np.frombuffer(seq, dtype=np.uint8)
        n_probes = (num_kmers + stride - 1) // stride
        n_coarse = (num_kmers + coarse_stride - 1) // coarse_stride
        words = np.zeros(n_probes + 1, dtype=np.uint32)
        decided = np.zeros(max(num_kmers, 1), dtype=np.uint32)
        periods = np.zeros(max(n_coarse, 1), dtype=np.uint32)
        coarse = np.zeros(max(n_coarse, 1), dtype=np.uint32)
        steps = self.L.hs_repeat_probes2(self.h, buf.ctypes.data, buf.size, num_kmers, kmin, kmax, stride, coarse_stride,
                                         words.ctypes.data, decided.ctypes.data, 1, periods.ctypes.data, coarse.ctypes.data)
        return words[:n_probes], decided[:num_kmers], int(steps), periods[:n_coarse], coarse[:n_coarse]

    def build_dict(self, seed_level: int, length: int) -> int:
        """the repeat dictionary of strings of `length` bases, derived from the seed level `seed_level` (as nm_build_dict)"""
        return int(self.L.hs_build_dict(self.h, seed_level, length))

    def dict_lookup(self, kmer: bytes) -> int:
        return int(self.L.hs_dict_lookup(self.h, kmer))

    def guard(self, seq: bytes, num_kmers, ks, is_range: bool, initial_len: int = 0, use_rc: bool = True):
        """nm_guard_range_one / nm_guard_list_one: first position of the segment for which the reference would raise, or None"""
        buf = np.frombuffer(seq, dtype=np.uint8)
        k = np.asarray([min(ks), max(ks)] if is_range else list(ks), dtype=np.uint32)
        p = int(self.L.hs_guard(self.h, buf.ctypes.data, buf.size, num_kmers, k.ctypes.data, k.size, int(is_range), initial_len, int(use_rc)))
        return None if p == 0xFFFFFFFFFFFFFFFF else p

    def sites(self, seq: bytes, num_kmers, kmin, kmax, d_cap=59, probes=1, ks=None, dtype=np.uint8, chance_max=256, walk_max=64):
        """k_sites -> gated repeat probes -> k_resolve, as the device runs them (needs check_quad() first: it builds the
        quad table).  ks: list mode (kmin / kmax are then its first / longest length).  Returns (elements, status,
        rc, need bitmap, counters) -- counters: table entries read, positions walked, probes run, probe-decided."""
        buf = np.frombuffer(seq, dtype=np.uint8)
        out = np.zeros(max(num_kmers, 1), dtype=dtype)
        status = np.zeros(8, dtype=np.uint64)
        need = np.zeros((num_kmers + 63) // 64 + 1, dtype=np.uint64)
        counters = np.zeros(5, dtype=np.uint64)
        k = np.asarray(ks if ks is not None else [], dtype=np.uint32)
        rc = self.L.hs_sites(self.h, buf.ctypes.data, buf.size, num_kmers, kmin, kmax, d_cap, probes,
                             k.ctypes.data if k.size else None, k.size, out.dtype.itemsize, out.ctypes.data,
                             status.ctypes.data, need.ctypes.data, counters.ctypes.data, chance_max, walk_max)
        return out[:num_kmers], status, rc, need[:-1], counters

    def fixed_k(self, seq: bytes, num_kmers, ks, use_rc=True, dtype=np.uint8):
        buf = np.frombuffer(seq, dtype=np.uint8)
        out = np.zeros(max(num_kmers, 1), dtype=dtype)
        status = np.zeros(8, dtype=np.uint64)
        k = np.asarray(ks, dtype=np.uint32)
        rc = self.L.hs_fixed_k(self.h, buf.ctypes.data, buf.size, num_kmers, k.ctypes.data, k.size, int(use_rc),
                               out.dtype.itemsize, out.ctypes.data, status.ctypes.data)
        return out[:num_kmers], status, rc

    def count(self, seq: bytes, starts, lens):
        buf = np.frombuffer(seq, dtype=np.uint8)
        s = np.ascontiguousarray(starts, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint64)
        out = np.zeros(s.size, dtype=np.uint32)
        self.L.hs_count(self.h, buf.ctypes.data, s.ctypes.data, l.ctypes.data, s.size, out.ctypes.data)
        return out

    @staticmethod
    def upper(seq: bytes, num_kmers, kmax):
        buf = np.frombuffer(seq, dtype=np.uint8)
        out = np.zeros(max(num_kmers, 1), dtype=np.uint32)
        lib().hs_upper(buf.ctypes.data, buf.size, num_kmers, kmax, out.ctypes.data)
        return out[:num_kmers]


def fasta_scan(path, threads=4, chunk_bytes=4 << 20, ranges=()):
    """newmap_amd/csrc/nm_fasta_scan.hpp (the FASTA scan of the native driver's parallel front-end) on a file:
    [(id bytes, data bytes)] of every record that holds data, as newmap_amd.fasta.fasta_records yields them; `ranges`
    = (record index among ALL scanned records, lo, hi) triples read back separately -> list of bytes."""
    L = lib()
    h = L.hs_fasta_open(str(path).encode(), threads, chunk_bytes)
    if not h:
        raise OSError(f"cannot scan {path}")
    try:
        out, pieces = [], []
        n = int(L.hs_fasta_count(h))
        lens = []
        for i in range(n):
            idbuf = ctypes.create_string_buffer(4096)
            nb = ctypes.c_uint64(0)
            idlen = int(L.hs_fasta_record(h, i, idbuf, 4096, ctypes.byref(nb)))
            lens.append(int(nb.value))
            if nb.value == 0:
                continue
            data = np.zeros(nb.value, dtype=np.uint8)
            L.hs_fasta_read(h, i, 0, nb.value, data.ctypes.data)
            out.append((idbuf.raw[:idlen], data.tobytes()))
        for i, lo, hi in ranges:
            piece = np.zeros(hi - lo, dtype=np.uint8)
            L.hs_fasta_read(h, i, lo, hi, piece.ctypes.data)
            pieces.append(piece.tobytes())
        return (out, lens, pieces) if ranges else out
    finally:
        L.hs_fasta_close(h)


def valid_bits(seq: bytes, kmin: int) -> np.ndarray:
    """nm_valid4 at every position: True when the kmin bases from there on are free of ambiguity"""
    buf = np.frombuffer(seq, dtype=np.uint8)
    out = np.zeros(buf.size, dtype=np.uint8)
    lib().hs_valid_bits(buf.ctypes.data, buf.size, kmin, out.ctypes.data)
    return out.astype(bool)


def check_codes4(rounds=200):
    """nm_base_codes4 against nm_base_code; returns the number of disagreements"""
    return int(lib().hs_check_codes4(rounds))


def multi(sims, seqs, num_kmers, kmin, kmax, ks=None, use_rc=True, dtype=np.uint8):
    """several sequences in lock-step x several indexes; ks=None: range mode"""
    L = lib()
    hs = (ctypes.c_void_p * len(sims))(*[s.h for s in sims])
    bufs = [np.frombuffer(s, dtype=np.uint8) for s in seqs]
    ptrs = (ctypes.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    out = np.zeros(max(num_kmers, 1), dtype=dtype)
    status = np.zeros(8, dtype=np.uint64)
    k = np.asarray(ks if ks is not None else [], dtype=np.uint32)
    rc = L.hs_multi(hs, len(sims), ptrs, len(bufs), len(seqs[0]), num_kmers, kmin, kmax,
                    k.ctypes.data if k.size else None, k.size, int(use_rc), out.dtype.itemsize,
                    out.ctypes.data, status.ctypes.data)
    return out[:num_kmers], status, rc

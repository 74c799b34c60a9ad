"""Python handle over the C-ABI (include/newmap_amd.h): one `Index` = one FM-index resident in
the HBM of one MI355X.  Plumbing only -- every count and every search runs in the HIP kernels
of csrc/nm_engine.hip."""
from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path
from typing import Sequence

from ._lazy import np        # numpy, imported at its first use: the one-shot CLI's native path never needs it (0.15 s)

from . import _lib


def default_device() -> int:
    """LOCAL_RANK when launched by torch.distributed.run, else NEWMAP_AMD_DEVICE, else 0."""
    for var in ("NEWMAP_AMD_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v not in (None, ""):
            return int(v)
    return 0


def device_count() -> int:
    return int(_lib.lib().nm_device_count())


def _as_u8(buf) -> np.ndarray:
    if isinstance(buf, np.ndarray):
        a = buf if buf.dtype == np.uint8 else buf.view(np.uint8)
        return np.ascontiguousarray(a)
    return np.frombuffer(memoryview(buf), dtype=np.uint8)


class Index:
    """Device-resident index.  Replaces the per-call load of src/newmap-count.c:9-17,135-136."""

    def __init__(self, index_path, device: int | None = None, seed_length: int | str | None = None, hbm_budget: int | None = None):
        """seed_length: None / "auto" = tables sized for throughput (seed ceil(log4 n)+2 <= 16 bases, quad table of
        cores as long as the free HBM allows: up to 180 GB), "auto-small" = the same kernels on tables of at
        most 20 GB (what the one-shot CLI uses), "file" = the --seed-length recorded by `newmap index`,
        0 = no seed table, 1..16 = that length.  hbm_budget (bytes): what the automatic sizes may take of the HBM for THIS
        index, all of it included -- the index files of one search share the device (nm_index_open_budget)."""
        self._L = _lib.lib()
        self.path = Path(index_path)
        self.device = default_device() if device is None else int(device)
        if seed_length is None:
            seed_length = os.environ.get("NEWMAP_AMD_SEED_LENGTH", "auto")
        code = {"auto": -2, "auto-small": -3, "file": -1}.get(seed_length)
        if code is None:
            code = int(seed_length)
        h = ctypes.c_void_p()
        if hbm_budget:
            rc = self._L.nm_index_open_budget(os.fsencode(self.path), self.device, code, int(hbm_budget), ctypes.byref(h))
        else:
            rc = self._L.nm_index_open(os.fsencode(self.path), self.device, code, ctypes.byref(h))
        _lib.raise_for(rc)
        self._h = h
        self._lock = threading.Lock()
        if os.environ.get("NEWMAP_AMD_KERNEL"):            # A/B measurements, see set_kernel
            self.set_kernel(int(os.environ["NEWMAP_AMD_KERNEL"]))
        if os.environ.get("NEWMAP_AMD_REPEAT_PROBES"):
            self.set_repeat_probes(os.environ["NEWMAP_AMD_REPEAT_PROBES"] != "0")
        if os.environ.get("NEWMAP_AMD_SITE_TABLE"):        # 1: the sites read the long cores, 2: the short ones (set_site_table)
            self.set_site_table(int(os.environ["NEWMAP_AMD_SITE_TABLE"]))
        if os.environ.get("NEWMAP_AMD_SEED_POLICY"):
            _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_SEED_POLICY,
                                                 int(os.environ["NEWMAP_AMD_SEED_POLICY"], 0)))

    # lifetime -----------------------------------------------------------------------------
    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.nm_index_close(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        if not self._h:
            raise ValueError("index is closed")
        return self._h

    def info(self) -> dict:
        names = ["bwt_length", "forward_text_length", "separators", "records", "raw_bases",
                 "seed_length", "device_bytes", "sa_ratio", "last_range_kernel", "pair_core_length", "device",
                 "lf_blocks", "two_step_blocks", "repeat_probes"]
        d = {n: int(self._L.nm_index_info(self.handle, i)) for i, n in enumerate(names)}
        d["quad_core_length"] = int(self._L.nm_index_info(self.handle, 18))
        d["quad_small_core_length"] = int(self._L.nm_index_info(self.handle, 19))
        d["last_site_core_length"] = int(self._L.nm_index_info(self.handle, 20))
        d["dict_length"] = int(self._L.nm_index_info(self.handle, 24))
        d["dict_entries"] = int(self._L.nm_index_info(self.handle, 25))
        d["lf2_blocks"] = int(self._L.nm_index_info(self.handle, 26))
        d["lcp_bytes"] = int(self._L.nm_index_info(self.handle, 35))
        return d

    def probe_tally(self) -> dict:
        """Repeat probes of the last range-mode launch: positions settled without a search and (with
        set_count_steps) the LF steps, rank blocks and seed entries the probes read."""
        names = ["lf_steps", "rank_blocks", "seed_lookups", "settled"]
        return {n: int(self._L.nm_index_info(self.handle, 14 + i)) for i, n in enumerate(names)}

    def open_words(self) -> dict:
        """k_open_words of the last launch: words of the need bitmap that hold open positions, and those positions, by class
        (>= 48, >= 24, >= 8, fewer open positions per word)"""
        v = [int(self._L.nm_index_info(self.handle, 27 + i)) for i in range(8)]
        return {"words": v[:4], "positions": v[4:]}

    def set_count_steps(self, on: bool):
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_COUNT_STEPS, int(bool(on))))

    def set_kernel(self, version: int):
        """0 = automatic, 1 = one lane per position (k_min_unique), 5 = the sites (k_sites + k_resolve)"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_KERNEL, int(version)))

    def set_repeat_probes(self, on: bool):
        """A/B: one probe per 64 positions settles stretches that occur twice over more than kmax bases
        (default on); off = every position searches for itself.  Results are identical."""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_REPEAT_PROBES, int(bool(on))))

    def set_list_via_range(self, on: bool):
        """A/B: list mode with one length on the range kernels (default) or on the list kernel."""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_LIST_VIA_RANGE, int(bool(on))))

    def set_lf_blocks(self, on: bool):
        """LF steps read one 16-byte LF entry (default) or the packed 32-byte rank block"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_LF_BLOCKS, int(bool(on))))

    def set_site_d(self, d_cap: int):
        """measurement / tests: cap on d = kmin - (quad core length + 4); a site settles a group of d + 5 positions"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_SITE_D, int(d_cap)))

    def set_site_table(self, which: int):
        """measurement / tests: the quad table the sites read: 0 = picked per launch, 1 = long cores, 2 = short cores"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_SITE_TABLE, int(which)))

    def set_lf2(self, on: bool):
        """A/B: walks take two bases per step (two-base LF blocks, where built) or one"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_LF2, int(bool(on))))

    def set_sweep(self, on: bool):
        """A/B: the positions the sites leave open are swept right to left where they are dense, neighbours sharing their walks
        (k_sweep; True = from the first launch on, 1 = the default: once the handle has met open positions) or each walks
        for itself (k_resolve; False)"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_SWEEP, 2 if on is True else int(on)))

    def set_lcp(self, on: bool):
        """A/B: where the end of a chain of the sweep moves, the index's LCP bytes give the new end (default, when the index
        file holds them) or the position walks for itself"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_LCP, int(bool(on))))

    def set_dictionary(self, on: bool):
        """A/B: open positions of the sites ask the repeat dictionary (default, when one was built and kmin allows) or the
        second quad table / the seed table"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_SEED_POLICY, 0 if on else 0x1000))

    def set_force_big(self, on: bool):
        """tests: exercise the code path of indexes beyond 2^31 positions on a small index"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_FORCE_BIG, int(bool(on))))

    def set_timing(self, on: bool):
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_TIMING, int(bool(on))))

    def read_timing(self, kind: int = 0):
        """(launches, total ms, longest ms) since the last read.  kind 0: the dominant search kernel of each segment
        alone; kind 1: all kernels of each segment (encode pass, sites, repeat probes, resolve)."""
        n, tot, mx = ctypes.c_uint64(0), ctypes.c_double(0), ctypes.c_double(0)
        _lib.raise_for(self._L.nm_timing_read_kind(self.handle, int(kind), ctypes.byref(n), ctypes.byref(tot), ctypes.byref(mx)))
        return int(n.value), float(tot.value), float(mx.value)

    # compat seam --------------------------------------------------------------------------
    def count_kmers(self, kmers: Sequence[bytes]) -> np.ndarray:
        n = len(kmers)
        offs = np.zeros(n + 1, dtype=np.uint64)
        if n:
            np.cumsum([len(k) for k in kmers], out=offs[1:])
        blob = _as_u8(b"".join(kmers)) if n else np.zeros(1, np.uint8)
        out = np.zeros(n, dtype=np.uint32)
        with self._lock:
            rc = self._L.nm_count_kmers(self.handle, blob.ctypes.data, offs.ctypes.data, n, out.ctypes.data)
        _lib.raise_for(rc)
        return out

    def count_from_sequence(self, seq, starts, lens) -> np.ndarray:
        s = np.ascontiguousarray(starts, dtype=np.uint64)
        l = np.ascontiguousarray(lens, dtype=np.uint64)
        if s.shape != l.shape:
            raise ValueError("Both lists of indices and lengths must be the same length")
        buf = _as_u8(seq)
        out = np.zeros(s.size, dtype=np.uint32)
        with self._lock:
            rc = self._L.nm_count_from_sequence(self.handle, buf.ctypes.data, buf.size, s.ctypes.data,
                                                l.ctypes.data, s.size, out.ctypes.data)
        if rc == _lib.NM_E_ARGUMENT:
            raise IndexError(_lib.last_error())          # src/newmap-count.c:184-190
        _lib.raise_for(rc)
        return out

    # fused hot path -----------------------------------------------------------------------
    def _not_found(self, seq: np.ndarray, bad_pos: int, length: int) -> str:
        kmer = bytes(seq[bad_pos:bad_pos + length]).decode("utf-8", "replace")
        return ("The following generated k-mer was not found in the index:\n"
                f"{kmer}\nPossibly a mismatch between the sequence and the index.")   # search.py:719-722

    def min_unique_segment(self, seq, num_kmers: int, kmin: int, kmax: int, use_revcomp: bool = True,
                           dtype=None, initial_search_length: int = 0):
        """newmap/search.py:383-548 for one segment -> (unique_lengths, n_ambiguous)."""
        if dtype is None:
            dtype = np.uint8 if kmax <= 255 else (np.uint16 if kmax <= 65535 else np.uint32)
        dtype = np.dtype(dtype)
        buf = _as_u8(seq)
        out = np.zeros(max(int(num_kmers), 1), dtype=dtype)
        amb, bad = ctypes.c_uint64(0), ctypes.c_uint64(0)
        with self._lock:
            rc = self._L.nm_min_unique_segment(self.handle, buf.ctypes.data, buf.size, int(num_kmers), int(kmin),
                                               int(kmax), int(initial_search_length), int(bool(use_revcomp)),
                                               dtype.itemsize, out.ctypes.data, ctypes.byref(amb),
                                               ctypes.byref(bad))
        _lib.raise_for(rc, self._not_found(buf, int(bad.value), kmin) if rc == _lib.NM_E_KMER_NOT_FOUND else None)
        return out[:num_kmers], int(amb.value)

    def fixed_k_segment(self, seq, num_kmers: int, kmer_lengths: Sequence[int], use_revcomp: bool = True,
                        dtype=None):
        """newmap/search.py:551-644 for one segment -> (unique_lengths, n_ambiguous)."""
        ks = np.ascontiguousarray(kmer_lengths, dtype=np.uint32)
        kmax = int(ks.max())
        if dtype is None:
            dtype = np.uint8 if kmax <= 255 else (np.uint16 if kmax <= 65535 else np.uint32)
        dtype = np.dtype(dtype)
        buf = _as_u8(seq)
        out = np.zeros(max(int(num_kmers), 1), dtype=dtype)
        amb, bad = ctypes.c_uint64(0), ctypes.c_uint64(0)
        with self._lock:
            rc = self._L.nm_fixed_k_segment(self.handle, buf.ctypes.data, buf.size, int(num_kmers),
                                            ks.ctypes.data, ks.size, int(bool(use_revcomp)), dtype.itemsize,
                                            out.ctypes.data, ctypes.byref(amb), ctypes.byref(bad))
        _lib.raise_for(rc, self._not_found(buf, int(bad.value), int(ks[0])) if rc == _lib.NM_E_KMER_NOT_FOUND else None)
        return out[:num_kmers], int(amb.value)

    def upper_bound_segment(self, seq, num_kmers: int, kmax: int) -> np.ndarray:
        """newmap/search.py:744-766 + :769-882 on the device."""
        buf = _as_u8(seq)
        out = np.zeros(max(int(num_kmers), 1), dtype=np.uint32)
        with self._lock:
            rc = self._L.nm_upper_bound_segment(self.handle, buf.ctypes.data, buf.size, int(num_kmers), int(kmax),
                                                out.ctypes.data)
        if rc == _lib.NM_E_ARGUMENT:
            raise AssertionError(_lib.last_error())      # search.py:780-784
        _lib.raise_for(rc)
        return out[:num_kmers]

    # record fingerprints and the exact guard (include/newmap_amd.h, csrc/nm_hash.h) ----------------
    def records(self):
        """(lengths, fingerprints) of the indexed records, sorted by (length, fingerprint)"""
        n = int(self._L.nm_index_records(self.handle, None, None, 0))
        lens, fps = np.zeros(max(n, 1), dtype=np.uint64), np.zeros(max(n, 1), dtype=np.uint64)
        self._L.nm_index_records(self.handle, lens.ctypes.data, fps.ctypes.data, n)
        return lens[:n], fps[:n]

    def has_record(self, length: int, fingerprint: int) -> bool:
        return bool(self._L.nm_index_has_record(self.handle, int(length), int(fingerprint) & 0xFFFFFFFFFFFFFFFF))

    def guard_segments(self) -> int:
        """segments that have gone through the exact guard on this handle (0 while every searched record is an indexed one)"""
        return int(self._L.nm_index_info(self.handle, 23))

    def last_fingerprint(self) -> int:
        """fingerprint of the positions of the last host-buffer segment call (min_unique_segment / fixed_k_segment)"""
        return int(self._L.nm_index_info(self.handle, 21))

    def guard_segment(self, seq, num_kmers: int, kmer_lengths: Sequence[int], is_range: bool, use_revcomp: bool = True,
                      initial_search_length: int = 0):
        """The exact zero-count check of newmap/search.py:699-722 over one segment: raises RuntimeError exactly when some
        k-mer of the reference's probe schedule is absent from the index."""
        ks = np.ascontiguousarray([min(kmer_lengths), max(kmer_lengths)] if is_range else list(kmer_lengths), dtype=np.uint32)
        buf = _as_u8(seq)
        bad = ctypes.c_uint64(0)
        with self._lock:
            rc = self._L.nm_guard_segment(self.handle, buf.ctypes.data, buf.size, int(num_kmers), ks.ctypes.data, ks.size,
                                          int(bool(is_range)), int(initial_search_length), int(bool(use_revcomp)), ctypes.byref(bad))
        _lib.raise_for(rc, self._not_found(buf, int(bad.value), int(ks.min())) if rc == _lib.NM_E_KMER_NOT_FOUND else None)

    # native driver ------------------------------------------------------------------------------
    def set_segment_guard(self, on: bool):
        """on (default): min_unique_segment / fixed_k_segment raise like the reference's binary_search / linear_search for ANY
        absent probe (the exact guard runs unless the segment is a whole indexed record); off: the caller checks whole
        records itself (newmap_amd/search.py's Python driver)"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_SEGMENT_GUARD, int(bool(on))))

    def set_initial_search_length(self, n: int):
        """--initial-search-length of the run: shapes the reference's probe schedule, which the exact guard replays"""
        _lib.raise_for(self._L.nm_set_option(self.handle, _lib.NM_OPT_INITIAL_LENGTH, int(n or 0)))

    def guard_fasta(self, fasta_path, kmer_lengths, is_range: bool, use_revcomp: bool, batch: int, include, exclude, flags):
        """nm_guard_fasta: the exact zero-count check over the records with flags[i] != 0 (records with data, file order)"""
        ks = np.ascontiguousarray([min(kmer_lengths), max(kmer_lengths)] if is_range else list(kmer_lengths), dtype=np.uint32)
        inc = (ctypes.c_char_p * max(len(include), 1))(*[bytes(x) for x in include])
        exc = (ctypes.c_char_p * max(len(exclude), 1))(*[bytes(x) for x in exclude])
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        with self._lock:
            rc = self._L.nm_guard_fasta(self.handle, os.fsencode(fasta_path), ks.ctypes.data, ks.size, int(bool(is_range)),
                                        int(bool(use_revcomp)), int(batch), inc, len(include), exc, len(exclude), fl.ctypes.data, fl.size)
        _lib.raise_for(rc)

    def search_fasta(self, fasta_path, out_dir, kmer_lengths, is_range: bool, use_revcomp: bool = True,
                     batch: int = 10_000_000, include=(), exclude=(), on_record=None, rank: int = 0, world: int = 1,
                     record_info: list | None = None):
        """nm_search_fasta(_shard): FASTA in -> `<id>.unique.<dtype>` files out, natively.  `on_record(id: bytes,
        summary: dict)` is called once per output file.  world > 1: this process is one rank of a job with one
        process per GPU and searches / writes only its own interleaved share; `record_info` (a list) then receives one
        (length, fingerprint of this rank's share, searched) per FASTA record with data -- the caller joins the ranks
        (newmap_amd/parallel.py).  With one rank the record check and the exact guard run inside the call.  Returns the
        totals as a dict."""
        lengths = [min(kmer_lengths), max(kmer_lengths)] if is_range else [int(k) for k in kmer_lengths]
        ks = (ctypes.c_uint32 * len(lengths))(*lengths)                     # (no numpy on this path: newmap_amd/_lazy.py)
        fields = [f[0] for f in _lib.SearchSummary._fields_]

        def _cb(rec_id, summary, _user):
            if on_record is not None:
                on_record(rec_id, {f: int(getattr(summary.contents, f)) for f in fields})

        cb = _lib.RECORD_CALLBACK(_cb)
        inc = (ctypes.c_char_p * max(len(include), 1))(*[bytes(x) for x in include])
        exc = (ctypes.c_char_p * max(len(exclude), 1))(*[bytes(x) for x in exclude])
        total = _lib.SearchSummary()
        cap = 1 << 20
        info = (ctypes.c_uint64 * (3 * cap if record_info is not None else 3))()
        n_info = ctypes.c_uint64(0)
        with self._lock:
            rc = self._L.nm_search_fasta_shard_ex(self.handle, os.fsencode(fasta_path), os.fsencode(out_dir), ctypes.addressof(ks),
                                                  len(lengths), int(bool(is_range)), int(bool(use_revcomp)), int(batch),
                                                  inc, len(include), exc, len(exclude), cb, None, ctypes.byref(total),
                                                  int(rank), int(world), ctypes.addressof(info) if record_info is not None else None,
                                                  cap if record_info is not None else 0, ctypes.byref(n_info))
        if record_info is not None:
            if n_info.value > cap:
                raise OverflowError(f"{n_info.value} FASTA records: more than the sharded search keeps fingerprints for")
            record_info.extend((int(info[3 * i]), int(info[3 * i + 1]), int(info[3 * i + 2])) for i in range(int(n_info.value)))
        if rc == _lib.NM_E_ARGUMENT:
            msg = _lib.last_error()
            if "nothing was processed" in msg or "included sequences" in msg:
                return {f: 0 for f in fields}             # the caller raises the reference's ValueError
        _lib.raise_for(rc)
        return {f: int(getattr(total, f)) for f in fields}

    # device-resident variants (bench, multi-GPU driver) ---------------------------------------
    def min_unique_segment_dev(self, d_seq: int, seq_len: int, num_kmers: int, kmin: int, kmax: int,
                               use_revcomp: bool, elem_bytes: int, d_out: int, d_status: int, stream: int = 0):
        """asynchronous on `stream` (0 = the handle's own).  Independent segments given on different streams overlap
        on the device: the handle keeps its launch scratch per stream (include/newmap_amd.h, "Streams")"""
        rc = self._L.nm_min_unique_segment_dev(self.handle, d_seq, seq_len, num_kmers, kmin, kmax,
                                               int(bool(use_revcomp)), elem_bytes, d_out, d_status, stream or None)
        _lib.raise_for(rc)

    def fixed_k_segment_dev(self, d_seq: int, seq_len: int, num_kmers: int, kmer_lengths, use_revcomp: bool,
                            elem_bytes: int, d_out: int, d_status: int, stream: int = 0):
        ks = np.ascontiguousarray(kmer_lengths, dtype=np.uint32)
        rc = self._L.nm_fixed_k_segment_dev(self.handle, d_seq, seq_len, num_kmers, ks.ctypes.data, ks.size,
                                            int(bool(use_revcomp)), elem_bytes, d_out, d_status, stream or None)
        _lib.raise_for(rc)


def fingerprint(seq) -> int:
    """csrc/nm_hash.h on the host (small inputs, tests)"""
    buf = _as_u8(seq)
    return int(_lib.lib().nm_fingerprint_sequence(buf.ctypes.data, buf.size))


def fingerprint_join(fp_a: int, len_a: int, fp_b: int) -> int:
    """fingerprint of a . b from that of a (len_a bases, a multiple of 64) and that of b"""
    if len_a % 64:
        raise ValueError("segments join at multiples of 64 bases of their record")
    return int(_lib.lib().nm_fingerprint_join(int(fp_a), int(len_a), int(fp_b)))


def search_segment_multi(indexes: Sequence[Index], seqs: Sequence, num_kmers: int, kmer_lengths: Sequence[int],
                         is_range: bool, use_revcomp: bool = True, dtype=None):
    """nm_search_segment_multi: segments of several FASTA files in lock-step against several indexes
    (newmap/search.py:251-265, 461, 656-697) -> (unique_lengths, n_ambiguous)."""
    L = _lib.lib()
    ks = np.ascontiguousarray([min(kmer_lengths), max(kmer_lengths)] if is_range else list(kmer_lengths), dtype=np.uint32)
    kmax = int(ks.max())
    if dtype is None:
        dtype = np.uint8 if kmax <= 255 else (np.uint16 if kmax <= 65535 else np.uint32)
    dtype = np.dtype(dtype)
    bufs = [_as_u8(s) for s in seqs]
    if len({b.size for b in bufs}) != 1:
        raise ValueError("the segments of all FASTA files must have the same length")
    hs = (ctypes.c_void_p * len(indexes))(*[ix.handle for ix in indexes])
    ptrs = (ctypes.c_void_p * len(bufs))(*[b.ctypes.data for b in bufs])
    out = np.zeros(max(int(num_kmers), 1), dtype=dtype)
    amb, bad = ctypes.c_uint64(0), ctypes.c_uint64(0)
    rc = L.nm_search_segment_multi(hs, len(indexes), ptrs, len(bufs), bufs[0].size, int(num_kmers), ks.ctypes.data,
                                   ks.size, int(bool(is_range)), int(bool(use_revcomp)), dtype.itemsize,
                                   out.ctypes.data, ctypes.byref(amb), ctypes.byref(bad))
    _lib.raise_for(rc, indexes[0]._not_found(bufs[0], int(bad.value), int(ks.min())) if rc == _lib.NM_E_KMER_NOT_FOUND else None)
    return out[:num_kmers], int(amb.value)


# ---------------------------------------------------------------------------------------------
# The reference's FFI is stateless by path (index loaded and freed inside every call).  Keep the
# same call shape but hold the uploaded index across calls.
_cache: dict[tuple, Index] = {}
_cache_lock = threading.Lock()


def cached_index(index_path, device: int | None = None, hbm_budget: int | None = None) -> Index:
    """the resident handle of an index file; hbm_budget: see Index (used when the file is opened, not for a cached handle)"""
    p = Path(index_path)
    try:
        st = p.stat()
    except OSError:
        raise OSError(f"Could not load reference index from file {index_path}") from None   # newmap-count.c:13-16
    dev = default_device() if device is None else device
    key = (str(p.resolve()), st.st_mtime_ns, st.st_size, dev)
    with _cache_lock:
        ix = _cache.get(key)
        if ix is None or not ix._h:
            for k in [k for k in _cache if k[0] == key[0] and k[3] == dev]:
                _cache.pop(k).close()
            ix = Index(p, dev, None, hbm_budget)
            _cache[key] = ix
        return ix


def cached_indexes(index_paths, device: int | None = None) -> list:
    """the resident handles of the index files of ONE search (newmap/search.py:656-697 sums the counts of every file): files
    that have to be opened share the HBM that is free now in equal budgets, so that each gets tables for its share instead
    of the first one taking all of it"""
    paths = list(index_paths)
    if len(paths) <= 1:
        return [cached_index(p, device) for p in paths]
    dev = default_device() if device is None else device
    free = int(_lib.lib().nm_dev_free_bytes(dev))
    budget = int(free * 0.9 / len(paths)) if free else None
    return [cached_index(p, device, budget) for p in paths]


def close_all():
    with _cache_lock:
        for ix in _cache.values():
            ix.close()
        _cache.clear()

"""Multi-GPU `search`: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

A position's result depends only on the bytes [p, p+kmax) of its own record and on the read-only
index, so positions shard with no data-path collective:

  * the index is replicated -- every rank opens the same index file and uploads it to its own HBM;
  * the concatenated position space of all selected records is cut into world x R equal chunks of ~64 M positions,
    chunk c owned by rank c % world (`interleaved_ranges`: equal shares, and a cluster of long repeats is spread over
    all ranks); a chunk is cut into work units of at most `kmer_batch_size` positions plus kmax-1 bytes of
    lookahead taken from the same record (the reference's own segment rule,
    newmap/search.py:229-235, newmap/fasta.py:109-150);
  * default: every rank runs the native driver on its own share (`nm_search_fasta_shard`): it strips only the byte
    ranges of its units out of the FASTA, searches them on its GPU and writes them straight into the per-record
    `<id>.unique.<dtype>` files at their offsets -- each GPU drains over its own PCIe link, nothing crosses xGMI and no
    rank holds the whole FASTA;
  * gzip input takes the same path (every rank inflates the file once into memory and strips its own ranges);
    jobs whose ranks do not share a file system (NEWMAP_AMD_GATHER=1): ONE collective at the end gathers the padded
    per-rank results on rank 0 (RCCL), which writes the files.

The same code runs on CPU tensors with the "gloo" backend (tests/test_parallel_gloo.py), with the
per-unit compute injected.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Sequence

import numpy as np


@dataclass(frozen=True)
class Unit:
    record: int          # index into the record list
    start: int           # first position inside the record
    count: int           # positions in this unit
    seg_len: int         # bytes handed to the engine: count + lookahead, clipped at the record end


def shard_bounds(total: int, world: int) -> list[tuple[int, int]]:
    """contiguous, near-equal slices of [0, total)"""
    per = -(-total // world) if total else 0
    return [(min(r * per, total), min((r + 1) * per, total)) for r in range(world)]


def interleaved_ranges(total: int, world: int, target: int = 64 << 20) -> list[list[tuple[int, int]]]:
    """The global position space [0, total) cut into world x R equal chunks of about `target` positions, chunk c owned
    by rank c % world: every rank gets the same number of positions (to within R), and a stretch that is expensive to
    search -- a cluster of long repeats (BASELINE configs[4]) -- is spread over all ranks instead of landing on one
    ("dealt round-robin rather than whole chromosomes").  Returns the ranges of every rank."""
    if total <= 0:
        return [[] for _ in range(world)]
    rounds = max(1, -(-total // (world * max(int(target), 1))))
    n_chunks = world * rounds
    edges = [total * c // n_chunks for c in range(n_chunks + 1)]
    out: list[list[tuple[int, int]]] = [[] for _ in range(world)]
    for c in range(n_chunks):
        if edges[c + 1] > edges[c]:
            out[c % world].append((edges[c], edges[c + 1]))
    return out


def units_for_ranges(record_lengths: Sequence[int], ranges: Sequence[tuple[int, int]], batch: int, kmax: int) -> list[Unit]:
    """work units covering the given ranges of global positions, in order"""
    units: list[Unit] = []
    for lo, hi in ranges:
        units.extend(units_for_slice(record_lengths, lo, hi, batch, kmax))
    return units


def units_for_slice(record_lengths: Sequence[int], lo: int, hi: int, batch: int, kmax: int) -> list[Unit]:
    """work units covering global positions [lo, hi) (global = records laid end to end)"""
    units: list[Unit] = []
    base = 0
    for r, n in enumerate(record_lengths):
        a, b = max(lo, base), min(hi, base + n)
        p = a
        while p < b:
            cnt = min(batch, b - p)
            start = p - base
            seg_len = min(start + cnt + kmax - 1, n) - start
            units.append(Unit(r, start, cnt, seg_len))
            p += cnt
        base += n
    return units


def run_slice(records: Sequence[tuple[bytes, bytes]], units: Sequence[Unit],
              compute: Callable[..., np.ndarray], dtype, companions: Sequence[Sequence[bytes]] = ()) -> np.ndarray:
    """results of this rank's units, concatenated in global position order.  `companions`: the record data of further
    FASTA files searched in lock-step with `records` (newmap/search.py:251-265: geometry from the first file); with
    them `compute` is called with the LIST of the files' segments instead of one segment."""
    parts = []
    for u in units:
        data = records[u.record][1]
        seg = data[u.start:u.start + u.seg_len]
        if companions:
            seg = [seg] + [c[u.record][u.start:u.start + u.seg_len] for c in companions]
        arr = np.asarray(compute(seg, u.count), dtype=dtype)
        if arr.size != u.count:
            raise RuntimeError("engine returned a result of the wrong length")
        parts.append(arr)
    return np.concatenate(parts) if parts else np.zeros(0, dtype=dtype)


def _group_formed() -> bool:
    """a torch.distributed process group exists (a job launched by torch.distributed.run, even with ONE rank: its
    collectives then run over RCCL / gloo like those of a larger job)"""
    import sys
    if "torch" not in sys.modules:
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def gather_to_root(local: np.ndarray, total: int, world: int, rank: int, device=None) -> np.ndarray | None:
    """ONE collective: equal padded slices -> rank 0.  Returns the full array on rank 0."""
    if world == 1 and not _group_formed():
        return local
    import torch
    import torch.distributed as dist
    per = -(-total // world)
    dev = device if device is not None else torch.device("cpu")
    raw = np.ascontiguousarray(local).view(np.uint8)
    item = local.dtype.itemsize
    buf = torch.zeros(per * item, dtype=torch.uint8, device=dev)
    if raw.size:
        buf[:raw.size].copy_(torch.from_numpy(raw.copy()))
    if rank == 0:
        out = [torch.empty(per * item, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(buf, out, dst=0)
        full = torch.cat(out).cpu().numpy().view(local.dtype)
        bounds = shard_bounds(total, world)
        pieces = [full[r * per:r * per + (hi - lo)] for r, (lo, hi) in enumerate(bounds)]
        return np.concatenate(pieces)
    dist.gather(buf, None, dst=0)
    return None


def file_layout(records: Sequence[tuple[bytes, bytes]]) -> list[tuple[bool, int, int]]:
    """Which file bytes a record owns, the reference's way (newmap/search.py:268-305): adjacent records with one id
    append to one file; an id that comes back later truncates the file again, so only its LAST run of records is kept.
    Returns per record (owns output, first element of the record inside its file, elements in the whole file)."""
    runs: list[list[int]] = []
    for i, (rid, _) in enumerate(records):
        if runs and records[runs[-1][0]][0] == rid:
            runs[-1].append(i)
        else:
            runs.append([i])
    last_run = {records[r[0]][0]: k for k, r in enumerate(runs)}
    out: list[tuple[bool, int, int]] = [(False, 0, 0)] * len(records)
    for k, r in enumerate(runs):
        owns = last_run[records[r[0]][0]] == k
        total = sum(len(records[i][1]) for i in r)
        off = 0
        for i in r:
            out[i] = (owns, off, total)
            off += len(records[i][1])
    return out


def write_ranges_direct(records: Sequence[tuple[bytes, bytes]], parts: Sequence[tuple[int, np.ndarray]],
                        path_of: Callable[[bytes], Path], rank: int, barrier: Callable[[], None]) -> None:
    """No collective: rank 0 creates every output file at its full length, then each rank writes its results --
    `parts` = (first global position, elements), global = records laid end to end -- at their byte offsets.  Records
    with one id: see `file_layout`."""
    import os
    layout = file_layout(records)
    if rank == 0:
        done = set()
        for (rid, _), (owns, _, total) in zip(records, layout):
            if owns and rid not in done:
                done.add(rid)
                with open(path_of(rid), "wb") as fh:
                    fh.truncate(total * (parts[0][1].dtype.itemsize if parts else 1))
    barrier()
    starts = np.concatenate(([0], np.cumsum([len(d) for _, d in records]))).astype(np.int64)
    for lo, arr in parts:
        item = arr.dtype.itemsize
        hi = lo + arr.size
        for i, (rid, data) in enumerate(records):
            a, b = max(lo, int(starts[i])), min(hi, int(starts[i + 1]))
            owns, off0, _ = layout[i]
            if a >= b or not owns:
                continue
            fd = os.open(path_of(rid), os.O_WRONLY)
            try:
                buf = memoryview(np.ascontiguousarray(arr[a - lo:b - lo])).cast("B")
                off = (off0 + a - int(starts[i])) * item
                while len(buf):
                    w = os.pwrite(fd, buf, off)
                    buf, off = buf[w:], off + w
            finally:
                os.close(fd)
    barrier()


def write_slice_direct(records: Sequence[tuple[bytes, bytes]], local: np.ndarray, lo: int, hi: int,
                       path_of: Callable[[bytes], Path], rank: int, barrier: Callable[[], None]) -> None:
    """one contiguous slice [lo, hi) of the position space (kept for callers with contiguous shares)"""
    if rank == 0 and local.size == 0:
        local = np.zeros(0, dtype=local.dtype)
    # (an empty share still takes part in the file creation and the barriers)
    parts = [(lo, local)] if hi > lo else []
    if rank == 0 and not parts:
        parts = [(0, np.zeros(0, dtype=local.dtype))]
    write_ranges_direct(records, parts, path_of, rank, barrier)


def search_records_sharded(records: Sequence[tuple[bytes, bytes]], compute: Callable[..., np.ndarray],
                           kmax: int, batch: int, dtype, world: int, rank: int, device=None,
                           companions: Sequence[Sequence[bytes]] = ()):
    """Returns {record id: array} on rank 0 (None elsewhere)."""
    lengths = [len(d) for _, d in records]
    total = int(sum(lengths))
    lo, hi = shard_bounds(total, world)[rank]
    units = units_for_slice(lengths, lo, hi, batch, kmax)
    local = run_slice(records, units, compute, dtype, companions)
    full = gather_to_root(local, total, world, rank, device)
    if full is None:
        return None
    out, base = {}, 0
    layout = file_layout(records)             # adjacent records with one id append; a later run of an id wins
    for (rid, _), n, (owns, off, total) in zip(records, lengths, layout):
        if owns:
            if off == 0:
                out[rid] = np.zeros(total, dtype=full.dtype)
            out[rid][off:off + n] = full[base:base + n]
        base += n
    return out


def _raise_together(world: int, rank: int, failure: Exception | None) -> None:
    """every rank learns whether any rank failed (one small all-reduce in the place of a bare barrier): the failing rank
    re-raises its own exception, the others raise a RuntimeError naming it -- nobody is left waiting in a collective"""
    if world <= 1 and not _group_formed():
        if failure is not None:
            raise failure
        return
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    bad = torch.tensor([rank + 1 if failure is not None else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if failure is not None:
        raise failure
    if int(bad.item()):
        raise RuntimeError(f"the search failed on rank {int(bad.item()) - 1} (see its message); the output files are incomplete")


def unverified_records(index, info, world: int) -> list[int]:
    """flags per FASTA record with data: 1 = searched but not one of the indexed records (exact guard needed).  `info` =
    this rank's (length, fingerprint of its share, searched) per record; the shares are added up over the ranks mod 2^64
    (as four 16-bit limbs each, so that no backend's integer sum can overflow)."""
    n = len(info)
    fps = np.array([fp for _, fp, _ in info], dtype=np.uint64)
    searched = np.array([s for _, _, s in info], dtype=np.int64)
    if (world > 1 or _group_formed()) and n:
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        limbs = np.stack([(fps >> np.uint64(16 * j)) & np.uint64(0xFFFF) for j in range(4)] + [searched.astype(np.uint64)]).astype(np.int64)
        t = torch.from_numpy(limbs).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        limbs = t.cpu().numpy().astype(np.uint64)
        with np.errstate(over="ignore"):
            fps = limbs[0] + (limbs[1] << np.uint64(16)) + (limbs[2] << np.uint64(32)) + (limbs[3] << np.uint64(48))
        # searched: 1 per rank for a searched record, 2 where a rank could not join its segments
        unjoinable = limbs[4] > np.uint64(world)
        searched = (limbs[4] > 0).astype(np.int64)
    else:
        unjoinable = searched == 2
    return [int(bool(searched[i]) and (bool(unjoinable[i]) or not index.has_record(info[i][0], int(fps[i])))) for i in range(n)]


def write_unique_counts_distributed(config) -> None:
    """`newmap search` over all ranks of the current torch.distributed job (or a single process).
    Output files are identical to newmap_amd.search.write_unique_counts."""
    import os
    from . import search as S
    from .engine import cached_index
    from .fasta import fasta_records
    from .util import optional_gzip_open

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = None
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ and "MASTER_PORT" in os.environ      # by torch.distributed.run, with any number of ranks
    if world > 1 or launched:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = os.environ.get("NEWMAP_AMD_DIST_BACKEND", "nccl")   # "gloo": rehearsals with several ranks on one GPU
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
        if dist.get_backend() == "nccl":
            device = torch.device("cuda", local_rank)

    kmax, kmin = max(config.kmer_lengths), min(config.kmer_lengths)
    dtype, suffix = S.output_type(kmax)
    # several FASTA files in lock-step and / or several index files (newmap/search.py:251-265, 656-697): the same plan --
    # units of the FIRST file's records -- with every file's segment of a unit handed to nm_search_segment_multi
    multi = len(config.fasta_filepaths) != 1 or len(config.fmindex_filepaths) != 1

    def barrier():
        if world > 1 or launched:
            import torch.distributed as dist
            dist.barrier()

    def nothing_found():
        if config.include_sequence_ids:
            raise ValueError(f"None of the included sequences were found: {config.include_sequence_ids}")
        raise ValueError("The excluded sequences were too strict and nothing was processed: "
                         f"{config.exclude_sequence_ids}")

    gather = (world > 1 or launched) and os.environ.get("NEWMAP_AMD_GATHER", "0") == "1"
    if not multi and not gather and os.environ.get("NEWMAP_AMD_PYTHON_DRIVER", "") != "1":
        # every rank runs the native driver on its own interleaved share of the position space
        index = cached_index(config.fmindex_filepaths[0], local_rank if config.device is None else config.device)
        index.set_initial_search_length(config.initial_search_length)
        info: list = []
        total, failure = None, None
        try:
            total = index.search_fasta(config.fasta_filepaths[0], config.output_directory, config.kmer_lengths,
                                       config.is_binary_search, config.use_reverse_complement, config.kmer_batch_size,
                                       config.include_sequence_ids, config.exclude_sequence_ids, None, rank, world, info)
        except Exception as e:                      # (a rank that fails still meets the others below: nobody waits for ever)
            failure = e
        _raise_together(world, rank, failure)
        # a record whose fingerprint -- the ranks' shares added up -- is in the index holds no absent k-mer; the others are
        # searched again by the exact guard on rank 0 (newmap/search.py:699-722; csrc/nm_hash.h)
        flags = unverified_records(index, info, world)
        failure = None
        # the records to guard are dealt over the ranks (every rank knows all flags): a multi-Gbp FASTA that is not the indexed
        # genome costs ~100 LF steps per position, and ranks waiting for ONE rank's guard would sit in the collective below
        # for longer than its watchdog allows
        order = 0
        mine = [0] * len(flags)
        for i, f in enumerate(flags):
            if f:
                mine[i] = int(order % max(world, 1) == rank)
                order += 1
        if any(mine):
            try:
                index.guard_fasta(config.fasta_filepaths[0], config.kmer_lengths, config.is_binary_search, config.use_reverse_complement,
                                  config.kmer_batch_size, config.include_sequence_ids, config.exclude_sequence_ids, mine)
            except Exception as e:
                failure = e
        _raise_together(world, rank, failure)
        if total["records"] == 0 and (config.include_sequence_ids or config.exclude_sequence_ids):
            nothing_found()
        return
    with optional_gzip_open(config.fasta_filepaths[0], "rb") as fh:
        every = list(fasta_records(fh))
    keep = [S._wanted(config, rid) for rid, _ in every]
    records = [r for r, k in zip(every, keep) if k]
    del every
    if not records:
        nothing_found()
    companions = []
    for path in config.fasta_filepaths[1:]:
        with optional_gzip_open(path, "rb") as fh:
            other = [data for rid, data in fasta_records(fh)]
        # (lock-step is by position in the file, newmap/search.py:260: record i of every file, whatever its id)
        other = [d for d, k in zip(other, keep) if k]
        if len(other) < len(records) or any(len(d) != len(r[1]) for d, r in zip(other, records)):
            raise ValueError(f"{path}: the sharded lock-step search needs the record lengths of {config.fasta_filepaths[0]} "
                             "(run a single process for files that differ)")
        companions.append(other[:len(records)])
    from .engine import cached_indexes
    indexes = cached_indexes(config.fmindex_filepaths, local_rank if config.device is None else config.device)
    index = indexes[0]

    def compute(seg, count: int) -> np.ndarray:
        if multi:
            from .engine import search_segment_multi
            segs = seg if isinstance(seg, list) else [seg]
            return search_segment_multi(indexes, segs, count, config.kmer_lengths, config.is_binary_search,
                                        config.use_reverse_complement, dtype)[0]
        if config.is_binary_search:
            return index.min_unique_segment(seg, count, kmin, kmax, config.use_reverse_complement, dtype)[0]
        return index.fixed_k_segment(seg, count, config.kmer_lengths, config.use_reverse_complement, dtype)[0]

    def path_of(rid: bytes) -> Path:
        return Path(config.output_directory) / S.UNIQUE_COUNT_FILENAME_FORMAT.format(rid.decode(), suffix)

    if gather:                                                          # ranks without a shared file system
        result = search_records_sharded(records, compute, kmax, config.kmer_batch_size, dtype, world, rank, device, companions)
        if result is not None:
            for rid, arr in result.items():
                with open(path_of(rid), "wb") as fh:
                    arr.tofile(fh)
        barrier()
        return
    lengths = [len(d) for _, d in records]
    parts = []
    for lo, hi in interleaved_ranges(int(sum(lengths)), world)[rank]:
        parts.append((lo, run_slice(records, units_for_slice(lengths, lo, hi, config.kmer_batch_size, kmax), compute, dtype, companions)))
    if not parts:
        parts = [(0, np.zeros(0, dtype=dtype))]
    write_ranges_direct(records, parts, path_of, rank, barrier)

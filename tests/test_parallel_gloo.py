"""The N>1 path on CPU: world_size-2, -3 and -8 gloo jobs run the sharding plan and the final
gather of newmap_amd.parallel with the per-unit compute injected (the oracle's closed form -- test
only; on a GPU box the compute is the HIP engine, covered by tests/test_gpu_parity.py)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_records():
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    a = bytearray(bytes(alpha[rng.integers(0, 4, 5000)]))
    a[1200:1300] = b"N" * 100
    b = bytes(alpha[rng.integers(0, 4, 1800)]) + bytes(a[100:600])
    c = bytes(alpha[rng.integers(0, 4, 37)])
    return [(b"a", bytes(a)), (b"b", b), (b"c", c)]


def _records_with_duplicates():
    r = _make_records()
    # (every record is a substring of an indexed one: the injected oracle raises on k-mers it has never seen)
    return [r[0], r[1], (b"b", r[0][1][2000:2700]), r[2], (b"a", r[1][1][:900])]


def _lock_step_inputs(records):
    """a second FASTA with the records' geometry and two indexes: of the first file and of another genome that shares a
    stretch with it.  (No total may fall strictly between 0 and the number of files -- the case the reference never
    terminates on, SURVEY A.3.7 -- so the second file repeats the first.)"""
    from oracle import ref_driver as rd
    rng = np.random.default_rng(5)
    second = [bytes(bytearray(d)) for _, d in records]          # (a copy: every total doubles, none falls between 0 and 2)
    other = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 900)]) + records[0][1][3000:3400]
    return second, [rd.OracleIndex([d for _, d in records]), rd.OracleIndex([other])]


def _single_process_files(records, compute, dtype):
    """what newmap_amd.search.write_unique_counts (and the reference, search.py:268-305) leaves on disk"""
    files = {}
    prev = None
    for rid, data in records:
        arr = np.asarray(compute(data), dtype=dtype)
        if rid == prev:
            files[rid] = np.concatenate((files[rid], arr))
        else:
            files[rid] = arr                      # truncate on a new id
        prev = rid
    return files


def _worker(rank, world, port, outdir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from newmap_amd import parallel
    from oracle import ref_driver as rd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    records = _make_records()
    oracle = rd.OracleIndex([d for _, d in records])
    kmin, kmax, batch = 8, 40, 700

    def compute(seg: bytes, count: int):
        # a unit is an epilogue-style segment whose lookahead may be shorter than kmax-1
        full = rd.closed_form_min_unique(seg, oracle, kmin, kmax)
        return full[:count]

    res = parallel.search_records_sharded(records, compute, kmax, batch, np.uint8, world, rank)
    if rank == 0:
        np.savez(os.path.join(outdir, f"w{world}.npz"), **{k.decode(): v for k, v in res.items()})
    else:
        assert res is None
    dist.barrier()
    # the direct output path: no collective, every rank writes its interleaved ranges into the shared files (uint16
    # here, so that byte offsets differ from positions).  Record ids the reference's way (search.py:268-305): "b" twice
    # in a row appends, "a" comes back later and its last run owns the file
    records2 = _records_with_duplicates()
    lengths = [len(d) for _, d in records2]
    parts = []
    for lo, hi in parallel.interleaved_ranges(sum(lengths), world, 1000)[rank]:
        parts.append((lo, parallel.run_slice(records2, parallel.units_for_slice(lengths, lo, hi, batch, kmax),
                                             lambda seg, count: rd.closed_form_min_unique(seg, oracle, kmin, kmax)[:count], np.uint16)))
    parallel.write_ranges_direct(records2, parts or [(0, np.zeros(0, np.uint16))], lambda rid: Path(outdir) / f"w{world}.{rid.decode()}.uint16",
                                 rank, dist.barrier)
    res2 = parallel.search_records_sharded(records2, compute, kmax, batch, np.uint8, world, rank)
    if rank == 0:
        np.savez(os.path.join(outdir, f"w{world}_dup.npz"), **{k.decode(): v for k, v in res2.items()})
    # two FASTA files in lock-step x two indexes (newmap/search.py:251-265, 656-697): the same plan, every file's segment
    # of a unit handed to the compute
    second, indexes = _lock_step_inputs(records)

    def compute_multi(segs, count):
        assert isinstance(segs, list) and len(segs) == 2 and len(segs[0]) == len(segs[1])
        arr, _ = rd.binary_search_segments_multi(indexes, [rd.Segment(b"r", s_, True) for s_ in segs], kmin, kmax, np.uint8)
        return arr[:count]

    res3 = parallel.search_records_sharded(records, compute_multi, kmax, batch, np.uint8, world, rank, None, [second])
    if rank == 0:
        np.savez(os.path.join(outdir, f"w{world}_multi.npz"), **{k.decode(): v for k, v in res3.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_search_equals_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    from oracle import ref_driver as rd
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / f"w{world}.npz")
    records = _make_records()
    oracle = rd.OracleIndex([d for _, d in records])
    for rid, data in records:
        want = rd.closed_form_min_unique(data, oracle, 8, 40)
        assert np.array_equal(got[rid.decode()], want), rid
    # lock-step files x several indexes: the reference-shaped single-process loop over whole records
    second, indexes = _lock_step_inputs(records)
    multi = np.load(tmp_path / f"w{world}_multi.npz")
    for (rid, data), other in zip(records, second):
        want, _ = rd.binary_search_segments_multi(indexes, [rd.Segment(rid, data, True), rd.Segment(rid, other, True)], 8, 40, np.uint8)
        assert np.array_equal(multi[rid.decode()], want), rid
    assert any((multi[rid.decode()] != got[rid.decode()]).any() for rid, _ in records)      # (the mode changes results)
    # files written without a collective, and the gathered result, for records that share ids
    want_files = _single_process_files(_records_with_duplicates(), lambda d: rd.closed_form_min_unique(d, oracle, 8, 40), np.uint16)
    dup = np.load(tmp_path / f"w{world}_dup.npz")
    assert sorted(dup.files) == sorted(k.decode() for k in want_files)
    for rid, want in want_files.items():
        assert np.array_equal(np.fromfile(tmp_path / f"w{world}.{rid.decode()}.uint16", dtype=np.uint16), want), rid
        assert np.array_equal(dup[rid.decode()], want.astype(np.uint8)), rid


def test_plan_covers_every_position_once():
    from newmap_amd import parallel
    lengths = [5000, 2300, 37, 1, 0, 999]
    total = sum(lengths)
    for world in (1, 2, 3, 8):
        seen = np.zeros(total, dtype=np.int32)
        for lo, hi in parallel.shard_bounds(total, world):
            for u in parallel.units_for_slice(lengths, lo, hi, 700, 40):
                base = sum(lengths[:u.record])
                seen[base + u.start: base + u.start + u.count] += 1
                assert u.count <= 700 and u.seg_len >= u.count
                assert u.start + u.seg_len <= lengths[u.record]
                assert u.seg_len == min(u.count + 39, lengths[u.record] - u.start)
        assert (seen == 1).all()


def test_interleaved_ranges_cover_and_balance():
    """the plan of the N > 1 jobs: every position exactly once, equal shares, chunks dealt round-robin"""
    from newmap_amd import parallel
    for total, world, target in ((3_088_269_832, 8, 64 << 20), (100_000_000, 8, 64 << 20), (1003, 3, 100), (5, 8, 64), (0, 4, 10)):
        per_rank = parallel.interleaved_ranges(total, world, target)
        assert len(per_rank) == world
        flat = sorted(r for rs in per_rank for r in rs)
        assert sum(hi - lo for lo, hi in flat) == total
        assert all(flat[i][1] == flat[i + 1][0] for i in range(len(flat) - 1)) and (not flat or (flat[0][0] == 0 and flat[-1][1] == total))
        sizes = [sum(hi - lo for lo, hi in rs) for rs in per_rank]
        if total >= world:
            assert max(sizes) - min(sizes) <= max(len(rs) for rs in per_rank)
        if total > world * target:
            assert all(len(rs) >= 2 for rs in per_rank)          # several chunks per rank: interleaved


def _tandem_spans(n: int, seed: int, frac: float = 0.5):
    """the tandem arrays newmap_amd.synth.tandem_dna(n, seed) lays down: same control flow, same draws from the
    generator (the bases are drawn and thrown away), no sequence kept"""
    rng = np.random.default_rng(seed)
    spans, pos, in_arrays = [], 0, 0
    while pos < n:
        m = min(int(rng.integers(50, 5001)), n - pos)
        rng.integers(0, 4, m, dtype=np.uint8)
        pos += m
        if pos >= n:
            break
        if in_arrays < frac * pos:
            rng.integers(0, 4, int(rng.integers(2, 201)), dtype=np.uint8)
            m = min(int(rng.integers(200, 50001)), n - pos)
            spans.append((pos, pos + m))
            pos += m
            in_arrays += m
    return np.array(spans, dtype=np.int64)


def test_interleaved_plan_spreads_repeat_clusters():
    """BASELINE configs[4] at full size (1 Gbp, 50 % tandem repeats, seed 20260517): a position inside an array costs a
    multiple of one outside (open after the sites, probe walks of up to kmax + 63 steps).  With the interleaved chunks of
    the N > 1 plan the per-rank cost differs by less than 10 %."""
    from newmap_amd import parallel, synth
    n = 1_000_000_000
    small = synth.tandem_dna(200_000, 20260517)               # the span model follows the generator exactly
    sp = _tandem_spans(200_000, 20260517)
    for a, b in sp[:5]:
        unit = None
        for period in range(2, 201):
            if b - a > 2 * period and (small[a + period:b] == small[a:b - period]).all():
                unit = period
                break
        assert unit is not None
    spans = _tandem_spans(n, 20260517)
    covered = np.concatenate(([0], np.cumsum(spans[:, 1] - spans[:, 0])))
    assert 0.45 < covered[-1] / n < 0.55

    def repeat_positions(lo, hi):                            # positions of [lo, hi) inside arrays
        def upto(x):
            k = int(np.searchsorted(spans[:, 0], x, side="right"))
            return int(covered[k]) - (max(int(spans[k - 1, 1]) - x, 0) if k else 0)
        return upto(hi) - upto(lo)

    for world in (2, 4, 8):
        per_rank = parallel.interleaved_ranges(n, world)
        loads = np.array([sum((hi - lo) + 20.0 * repeat_positions(lo, hi) for lo, hi in rs) for rs in per_rank])
        assert (loads.max() - loads.min()) / loads.mean() < 0.10, (world, loads)
        assert sum(repeat_positions(lo, hi) for rs in per_rank for lo, hi in rs) == covered[-1]

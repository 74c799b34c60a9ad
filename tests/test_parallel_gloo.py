"""The N>1 path on CPU: world_size-2 (and 3) gloo jobs run the sharding plan and the final
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
    # the default output path: no collective, every rank writes its slice into the shared files (uint16 here, so
    # that byte offsets differ from positions); records "a" twice: the later one owns the file
    records2 = records + [(b"a", records[1][1][:900])]
    lengths = [len(d) for _, d in records2]
    lo, hi = parallel.shard_bounds(sum(lengths), world)[rank]
    local = parallel.run_slice(records2, parallel.units_for_slice(lengths, lo, hi, batch, kmax),
                               lambda seg, count: rd.closed_form_min_unique(seg, oracle, kmin, kmax)[:count], np.uint16)
    parallel.write_slice_direct(records2, local, lo, hi, lambda rid: Path(outdir) / f"w{world}.{rid.decode()}.uint16",
                                rank, dist.barrier)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
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
    for rid, data in [records[1], records[2], (b"a", records[1][1][:900])]:        # files written without a collective
        want = rd.closed_form_min_unique(data, oracle, 8, 40).astype(np.uint16)
        assert np.array_equal(np.fromfile(tmp_path / f"w{world}.{rid.decode()}.uint16", dtype=np.uint16), want), rid


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


def _gather_worker(rank, world, port, outdir):
    """the double-buffered asynchronous gather of bench.py's N>1 step, on CPU tensors"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, steps = 1003, 5
    per = -(-n // world)
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    gather_bufs = [torch.empty(per * world, dtype=torch.uint8) for _ in range(2)] if rank == 0 else None
    pads = [torch.zeros(per, dtype=torch.uint8) for _ in range(2)]
    pending, seen = None, []
    for i in range(steps):
        d_out = ((torch.arange(n) * 7 + i) % 251).to(torch.uint8)          # this pass's full result
        b = i & 1
        pads[b][:hi - lo].copy_(d_out[lo:hi])
        if pending is not None:
            pending.wait()
            if rank == 0:
                seen.append(gather_bufs[(i - 1) & 1].clone())
        pending = dist.gather(pads[b], list(gather_bufs[b].split(per)) if rank == 0 else None, dst=0, async_op=True)
    pending.wait()
    if rank == 0:
        seen.append(gather_bufs[(steps - 1) & 1].clone())
        for i, g in enumerate(seen):
            want = ((torch.arange(n) * 7 + i) % 251).to(torch.uint8)
            got = torch.cat([g[r * per: r * per + (min((r + 1) * per, n) - min(r * per, n))] for r in range(world)])
            assert torch.equal(got, want), i
        Path(outdir, "gather_ok").write_text("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_async_double_buffered_gather_pattern(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_gather_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert (tmp_path / "gather_ok").exists()

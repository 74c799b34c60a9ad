"""Would consecutive segments gain from running on two lanes at once?  Experiment: the same segments of a bench
workload launched (a) one after the other on one handle and one stream, (b) alternately on TWO handles of the same
index with a stream each (every handle has its own scratch, so neighbouring segments may overlap on the GPU).

    python tools/two_lane_probe.py --config c2 --batch 10000000 [--steps 20]
    python tools/two_lane_probe.py --config c5 --batch 100000000 --steps 3
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from newmap_amd import _lib  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch
    import bench
    from newmap_amd import parallel
    from newmap_amd.engine import Index
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--batch", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--tables", default="auto-small")
    ap.add_argument("--graph", action="store_true", help="also capture a pass into a HIP graph (torch.cuda.CUDAGraph) and replay it")
    a = ap.parse_args()
    args = bench.parse(["--config", a.config, "--batch", str(a.batch)])
    wl = bench.headline_workload(args)
    fa, idx, _ = bench.prepare_index(args, wl, 0, lambda: None)
    dev = torch.device("cuda:0")
    KMIN, KMAX = wl.krange
    units = parallel.units_for_ranges(wl.lengths, [(0, wl.total)], a.batch, KMAX)
    handles = [Index(idx, 0, a.tables), Index(idx, 0, a.tables)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    seg_off = np.concatenate(([0], np.cumsum([(u.seg_len + 15) // 16 * 16 for u in units]))).astype(np.int64)
    out_off = np.concatenate(([0], np.cumsum([(u.count + 15) // 16 * 16 for u in units]))).astype(np.int64)
    d_seq = torch.empty(int(seg_off[-1]), dtype=torch.uint8, device=dev)
    d_out = [torch.zeros(int(out_off[-1]), dtype=torch.uint8, device=dev) for _ in range(2)]
    d_status = torch.zeros((len(units), _lib.NM_STATUS_WORDS), dtype=torch.int64, device=dev)
    for u, o in zip(units, seg_off[:-1]):
        d_seq[int(o):int(o) + u.seg_len].copy_(torch.from_numpy(wl.record(u.record)[u.start:u.start + u.seg_len]))
    torch.cuda.synchronize()

    def run(lanes, out):
        sp, op, st = d_seq.data_ptr(), out.data_ptr(), d_status.data_ptr()
        for i, u in enumerate(units):
            lane = i % lanes
            handles[lane].min_unique_segment_dev(sp + int(seg_off[i]), u.seg_len, u.count, KMIN, KMAX, True, 1,
                                                 op + int(out_off[i]), st + 8 * _lib.NM_STATUS_WORDS * i, streams[lane].cuda_stream)

    res = {"config": a.config, "batch": a.batch, "segments": len(units), "tables": a.tables}
    for lanes in (1, 2, 1, 2):
        for _ in range(2):
            run(lanes, d_out[lanes - 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            run(lanes, d_out[lanes - 1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        res.setdefault(f"lanes{lanes}_ms", []).append(round(dt * 1e3, 4))
        res.setdefault(f"lanes{lanes}_gpos", []).append(round(wl.total / dt / 1e9, 2))
    res["identical"] = bool(torch.equal(d_out[0], d_out[1]))
    # how long does the host take to ISSUE a pass (before any synchronize)?
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run(2, d_out[1])
    res["issue_ms_per_pass"] = round((time.perf_counter() - t0) / a.steps * 1e3, 4)
    torch.cuda.synchronize()
    if a.graph:
        # the same pass on ONE handle (lanes by stream) captured into a HIP graph and replayed
        ix = handles[0]
        main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream()
        out_g = torch.zeros_like(d_out[0])

        def one_pass(out, s0, s1):
            sp, op, st = d_seq.data_ptr(), out.data_ptr(), d_status.data_ptr()
            for i, u in enumerate(units):
                ix.min_unique_segment_dev(sp + int(seg_off[i]), u.seg_len, u.count, KMIN, KMAX, True, 1,
                                          op + int(out_off[i]), st + 8 * _lib.NM_STATUS_WORDS * i, (s0 if i % 2 == 0 else s1).cuda_stream)
        for _ in range(3):                      # buffers of both lanes grown, latches settled
            one_pass(out_g, main_s, side_s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(main_s):
            g.capture_begin()
            ev = torch.cuda.Event()
            ev.record(main_s)
            side_s.wait_event(ev)               # the second stream joins the capture
            one_pass(out_g, main_s, side_s)
            ev2 = torch.cuda.Event()
            ev2.record(side_s)
            main_s.wait_event(ev2)
            g.capture_end()
        torch.cuda.synchronize()
        out_g.zero_()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        res["graph_ms"] = round(dt * 1e3, 4)
        res["graph_gpos"] = round(wl.total / dt / 1e9, 2)
        res["graph_identical"] = bool(torch.equal(out_g, d_out[0]))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""End-to-end timings that bench.py deliberately leaves out (metric ii):
  * `newmap index` and `newmap search` wall time, FASTA in -> *.unique.uint8 files out (CLI, one GPU);
  * PCIe-inclusive rate of the host-buffer C-ABI call (nm_min_unique_segment: H2D + kernels + D2H).
Usage: python tools/e2e_timing.py [--mbp 100] [--out gpurun_out/e2e.json]"""
import argparse
import json
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=100.0)
    ap.add_argument("--config", default="c2", help="bench genome: c2 (uniform, --mbp), c3 (3.09 Gbp in 24 records), c5")
    ap.add_argument("--device-index", action="store_true", help="`newmap index --device 0` (suffix sort on the GPU)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from newmap_amd import synth
    from newmap_amd.engine import Index
    res = {"mbp": args.mbp}
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        recs = synth.config_genome(args.config, args.mbp if args.config == "c2" else None)
        (kmin, kmax) = {"c2": (20, 200), "c3": (24, 150), "c5": (20, 255)}[args.config]
        fa = synth.write_fasta(td / "genome.fa", recs)
        idx = td / "genome.awfmi"
        t0 = time.time()
        subprocess.run([sys.executable, "-m", "newmap_amd.main", "index"] + (["--device", "0"] if args.device_index else []) +
                       [str(fa), "-i", str(idx)], check=True, cwd=ROOT)
        res["cli_index_s"] = time.time() - t0
        out = td / "out"
        t0 = time.time()
        subprocess.run([sys.executable, "-m", "newmap_amd.main", "search", str(fa), str(idx), "-o", str(out),
                        "--search-range", f"{kmin}:{kmax}"], check=True, cwd=ROOT)
        res["cli_search_s"] = time.time() - t0
        n = recs[0][1].size
        res["positions"] = int(sum(r.size for _, r in recs))
        res["cli_search_positions_per_s"] = res["positions"] / res["cli_search_s"]
        got = np.fromfile(out / f"{recs[0][0]}.unique.uint8", dtype=np.uint8)
        assert got.size == n
        # host-buffer API, PCIe included: 10 M positions per call like the CLI
        seq = recs[0][1]
        with Index(idx, 0) as ix:
            ix.set_segment_guard(False)         # (pieces of a record: the caller of the seam checks whole records, as the drivers do)
            ix.min_unique_segment(seq[:10_000_000 + kmax - 1], 10_000_000, kmin, kmax)          # warm-up
            t0 = time.time()
            parts = []
            for p in range(0, n, 10_000_000):
                nk = min(10_000_000, n - p)
                seg = seq[p:min(p + nk + kmax - 1, n)]
                parts.append(ix.min_unique_segment(seg, nk, kmin, kmax)[0])
            dt = time.time() - t0
        res["host_api_positions_per_s"] = n / dt
        res["host_api_equals_cli_files"] = bool(np.array_equal(np.concatenate(parts), got))
        # the same C-ABI call with caller buffers that are REUSED (what a C host does): the Python wrapper above returns
        # a fresh array per call, and the first DMA into never-touched pages pays their faults and pinning
        import ctypes
        from newmap_amd import _lib
        L = _lib.lib()
        seg_out = np.zeros(10_000_000, dtype=np.uint8)
        seq_c = np.ascontiguousarray(seq)
        amb, bad = ctypes.c_uint64(0), ctypes.c_uint64(0)
        with Index(idx, 0) as ix:
            ix.set_segment_guard(False)
            same = True
            for rep in range(3):
                t0 = time.time()
                for p in range(0, n - n % 10_000_000, 10_000_000):
                    m = min(10_000_000 + kmax - 1, n - p)
                    rc = L.nm_min_unique_segment(ix.handle, seq_c.ctypes.data + p, m, 10_000_000, kmin, kmax, 0, 1, 1,
                                                 seg_out.ctypes.data, ctypes.byref(amb), ctypes.byref(bad))
                    assert rc == 0
                    if rep == 2:
                        same = same and bool(np.array_equal(seg_out, got[p:p + 10_000_000]))
                if rep == 1:
                    dt2 = time.time() - t0            # (second pass: every page touched before, no comparison inside)
            res["host_api_reused_buffers_positions_per_s"] = (n - n % 10_000_000) / dt2
            res["host_api_reused_buffers_equal_cli_files"] = same
    print(json.dumps(res))
    if args.out:
        Path(args.out).write_text(json.dumps(res))


if __name__ == "__main__":
    main()

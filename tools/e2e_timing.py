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
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from newmap_amd import synth
    from newmap_amd.engine import Index
    res = {"mbp": args.mbp}
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        recs = synth.config_genome("c2", args.mbp)
        fa = synth.write_fasta(td / "genome.fa", recs)
        idx = td / "genome.awfmi"
        t0 = time.time()
        subprocess.run([sys.executable, "-m", "newmap_amd.main", "index", str(fa), "-i", str(idx)], check=True, cwd=ROOT)
        res["cli_index_s"] = time.time() - t0
        out = td / "out"
        t0 = time.time()
        subprocess.run([sys.executable, "-m", "newmap_amd.main", "search", str(fa), str(idx), "-o", str(out),
                        "--search-range", "20:200"], check=True, cwd=ROOT)
        res["cli_search_s"] = time.time() - t0
        n = recs[0][1].size
        res["cli_search_positions_per_s"] = n / res["cli_search_s"]
        got = np.fromfile(out / "chr1.unique.uint8", dtype=np.uint8)
        assert got.size == n
        # host-buffer API, PCIe included: 10 M positions per call like the CLI
        seq = recs[0][1]
        with Index(idx, 0) as ix:
            ix.min_unique_segment(seq[:10_000_199], 10_000_000, 20, 200)          # warm-up
            t0 = time.time()
            parts = []
            for p in range(0, n, 10_000_000):
                nk = min(10_000_000, n - p)
                seg = seq[p:min(p + nk + 199, n)]
                parts.append(ix.min_unique_segment(seg, nk, 20, 200)[0])
            dt = time.time() - t0
        res["host_api_positions_per_s"] = n / dt
        res["host_api_equals_cli_files"] = bool(np.array_equal(np.concatenate(parts), got))
    print(json.dumps(res))
    if args.out:
        Path(args.out).write_text(json.dumps(res))


if __name__ == "__main__":
    main()

"""End-to-end time of the native driver (FASTA in -> <id>.unique.uint8 files out) on the bench genome for several numbers of
workers / slots, with the driver's own phase times (NEWMAP_AMD_DRIVER_TIMING).  Run after bench.py (it reuses its workdir).

    python tools/driver_sweep.py [--workers 8 12 16 24] [--dir /tmp/newmap_amd_bench/ns_3088.27mbp_device]
"""
import argparse
import json
import os
import shutil
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, nargs="+", default=[8, 12, 16, 24])
    ap.add_argument("--dir", default="/tmp/newmap_amd_bench/ns_3088.27mbp_device")
    ap.add_argument("--batch", type=int, default=10_000_000)
    a = ap.parse_args()
    from newmap_amd.engine import Index
    wd = Path(a.dir)
    res = []
    with Index(wd / "genome.awfmi", 0, "auto-small") as ix:
        for w in a.workers:
            os.environ["NEWMAP_AMD_DRIVER_SLOTS"] = str(w)
            os.environ["NEWMAP_AMD_DRIVER_TIMING"] = "1"
            best = None
            for rep in range(3):
                out = wd / "sweep_out"
                shutil.rmtree(out, ignore_errors=True)
                out.mkdir()
                t0 = time.time()
                total = ix.search_fasta(wd / "genome.fa", out, [20, 200], True, True, a.batch)
                dt = time.time() - t0
                best = dt if best is None or dt < best else best
            res.append({"workers": w, "batch": a.batch, "best_s": best, "positions_per_s": total["positions"] / best})
            print(json.dumps(res[-1]), flush=True)
            shutil.rmtree(out, ignore_errors=True)


if __name__ == "__main__":
    main()

"""Where the one-shot `newmap search` PROCESS spends its wall time on the 3.09 Gbp genome: the CLI run as a child with
NEWMAP_AMD_VERBOSE / NEWMAP_AMD_DRIVER_TIMING (phases of nm_index_open and of the driver on stderr) and Python-side time
stamps (interpreter + imports, main(), exit).  Reuses bench.py's workdir (run bench.py first).

    python tools/cli_breakdown.py [--dir /tmp/newmap_amd_bench/ns_3088.27mbp_device]
"""
import argparse
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r"""
import sys, time
t0 = time.time()
import newmap_amd.main as m
t1 = time.time()
m.parse_subcommands(sys.argv[1:])
t2 = time.time()
print(f"[child] imports {t1 - t0:.3f}s, main() {t2 - t1:.3f}s, numpy imported: {'numpy' in sys.modules}", file=sys.stderr, flush=True)
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/tmp/newmap_amd_bench/ns_3088.27mbp_device")
    a = ap.parse_args()
    wd = Path(a.dir)
    env = dict(os.environ, NEWMAP_AMD_VERBOSE="1", NEWMAP_AMD_DRIVER_TIMING="1")
    for rep in range(2):
        out = wd / "cli_out"
        shutil.rmtree(out, ignore_errors=True)
        t0 = time.time()
        p = subprocess.run([sys.executable, "-c", CHILD, "search", str(wd / "genome.fa"), str(wd / "genome.awfmi"), "-o", str(out),
                            "--search-range", "20:200"], cwd=ROOT, env=env, stderr=subprocess.PIPE, text=True)
        wall = time.time() - t0
        print(f"--- run {rep}: process wall {wall:.3f}s, rc {p.returncode}")
        for line in p.stderr.splitlines():
            if line.startswith(("[open]", "[driver]", "[child]", "[segment]")):
                print("   ", line)
        shutil.rmtree(out, ignore_errors=True)


if __name__ == "__main__":
    main()

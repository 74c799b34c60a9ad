"""Summarise rocprofv3 --pmc CSV output (one or more passes) into the per-dispatch means of one kernel.

    python tools/pmc_summary.py KERNEL_SUBSTRING OUT.csv BENCH.json PASS_DIR [PASS_DIR ...]

Each PASS_DIR holds the *_counter_collection.csv of one `rocprofv3 --pmc ... --output-format csv` run
(counters are collected in separate passes, as the MI355X guide prescribes).  Only dispatches of the
non-counting build of the kernel (template argument STATS = false) are averaged.  The first line of OUT.csv
ties the numbers to what produced them: the sha256 of the kernel sources (nm_kernels.hip.h + nm_core.h), the git
commit, the launch size and the quad table the sites read -- bench.py reports `traffic` only when they match
the run it is in."""
import collections
import csv
import glob
import hashlib
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    kernel, out, bench_json = sys.argv[1], sys.argv[2], sys.argv[3]
    sums = collections.OrderedDict()
    for d in sys.argv[4:]:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row["Kernel_Name"]
                    if kernel not in name or re.search(r"<(true|false), true[,>]", name):    # STATS builds
                        continue
                    per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for c, by_dispatch in per.items():
                vals = list(by_dispatch.values())
                sums[c] = (sum(vals) / len(vals), len(vals))
    h = hashlib.sha256()
    for f in ("newmap_amd/csrc/nm_kernels.hip.h", "newmap_amd/csrc/nm_core.h"):
        h.update((ROOT / f).read_bytes())
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "n/a (snapshot without .git)"
    except OSError:
        commit = "n/a"
    meta = {"kernel": kernel, "source_sha256": h.hexdigest()[:16], "git_commit": commit}
    try:
        r = json.loads(Path(bench_json).read_text().strip().splitlines()[-1])
        meta["workload"] = r["config"]["workload"].split(":")[0]
        meta["positions_per_launch"] = r["config"]["positions_this_rank"] // max(r["config"]["launches_per_step_per_rank"], 1)
        meta["site_core_length"] = r["roofline"].get("site_core_length", 0)
        meta["avg_launch_ms_hip_events"] = round(r["roofline"]["avg_launch_ms"], 5)
    except (OSError, ValueError, KeyError, IndexError):
        pass
    with open(out, "w") as fh:
        fh.write("# " + ", ".join(f"{k}={v}" for k, v in meta.items()) + "\n")
        fh.write(f"counter,mean_per_dispatch ({kernel}; dispatches averaged: {next(iter(sums.values()))[1] if sums else 0})\n")
        for c, (v, _) in sums.items():
            fh.write(f"{c},{v:.1f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()

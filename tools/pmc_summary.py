"""Summarise rocprofv3 --pmc CSV output (one or more passes) into the per-dispatch means of one kernel.

    python tools/pmc_summary.py KERNEL_SUBSTRING OUT.csv PASS_DIR [PASS_DIR ...]

Each PASS_DIR holds the *_counter_collection.csv of one `rocprofv3 --pmc ... --output-format csv` run
(counters are collected in separate passes, as the MI355X guide prescribes).  Only dispatches of the
non-counting build of the kernel (template argument STATS = false) are averaged."""
import collections
import csv
import glob
import re
import sys


def main():
    kernel, out = sys.argv[1], sys.argv[2]
    sums = collections.OrderedDict()
    for d in sys.argv[3:]:
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
    with open(out, "w") as fh:
        fh.write(f"counter,mean_per_dispatch ({kernel}; dispatches averaged: {next(iter(sums.values()))[1] if sums else 0})\n")
        for c, (v, _) in sums.items():
            fh.write(f"{c},{v:.1f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()

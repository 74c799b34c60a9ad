#!/bin/bash
# third sweep of tools/gather_ceiling: what page locality inside a wave is worth -- the lanes of a wave take
# their lines from one window of 64 KiB / 2 MiB / 1 GiB instead of anywhere in a 32 GiB table -- with the UTCL1 counters beside
set -o pipefail
O=${1:-gpurun_out/gather3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $O
G=tools/_build/gather_ceiling
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $G tools/gather_ceiling.hip || exit 1
: > $O/sweep3.jsonl
for loads in 1 3; do
  for pol in 0 1; do
    for page in 0 30 21 16 12; do
      timeout -k 5 100 $G 32 128 $loads $pol 28 32 64 $page >> $O/sweep3.jsonl || { echo "gather $loads $pol $page failed"; exit 1; }
    done
  done
done
cat $O/sweep3.jsonl
: > $O/pmc_utcl1.txt
for page in 0 30 21 16; do
  timeout -k 10 120 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_p$page -o p -- $G 32 128 3 1 26 32 64 $page > $O/pmc_p$page.json 2> $O/pmc_p$page.log || { echo "pmc page $page failed"; tail -3 $O/pmc_p$page.log; continue; }
  python3 - "$O" "$page" >> $O/pmc_utcl1.txt <<'PY'
import csv, glob, sys, collections
o, page = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{o}/pmc_p{page}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_gather" in row["Kernel_Name"]:
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
line = {k: sum(v.values()) / max(1, len(v)) for k, v in per.items()}
print(f"page_log2={page} (32 GiB, 128-B slots, lane pairs, nt) per launch:", {k: round(v) for k, v in sorted(line.items())})
PY
done
cat $O/pmc_utcl1.txt

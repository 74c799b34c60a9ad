#!/bin/bash
# second sweep of tools/gather_ceiling: one line read by ONE load instruction (1), by TWO instructions of a lane (2) or by
# TWO LANES of one instruction (3); then the request sizes of each form (PMC, at most 4 TCC counters per pass)
set -o pipefail
O=${1:-gpurun_out/gather2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $O
G=tools/_build/gather_ceiling
[ -x $G ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $G tools/gather_ceiling.hip || exit 1
: > $O/sweep2.jsonl
for gib in 2 32; do
  for cfg in "128 1 0 28 32" "128 2 0 28 32 64" "128 2 0 28 32 16" "128 2 0 28 32 32" "128 3 0 28 32 64" "128 3 0 28 32 16" "128 3 1 28 32 64" "128 3 2 28 32 64" "128 1 1 28 32" "64 3 0 28 32 16" "64 3 1 28 32 16"; do
    timeout -k 5 100 $G $gib $cfg >> $O/sweep2.jsonl || { echo "gather $gib $cfg failed"; exit 1; }
  done
done
cat $O/sweep2.jsonl
: > $O/pmc_request_sizes.txt
for cfg in "128 1 0" "128 2 0" "128 3 0" "128 3 1" "64 1 0" "64 1 1" "32 1 0" "32 1 1"; do
  tag=$(echo $cfg | tr ' ' _)
  timeout -k 10 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_$tag -o p -- $G 32 $cfg 26 32 > $O/pmc_$tag.json 2> $O/pmc_$tag.log || { echo "pmc $cfg failed"; tail -3 $O/pmc_$tag.log; continue; }
  timeout -k 10 120 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmcb_$tag -o p -- $G 32 $cfg 26 32 > $O/pmcb_$tag.json 2> $O/pmcb_$tag.log || echo "pmc-b $cfg failed"
  python3 - "$O" "$tag" "$cfg" >> $O/pmc_request_sizes.txt <<'PY'
import csv, glob, sys, collections
o, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for d in (f"{o}/pmc_{tag}", f"{o}/pmcb_{tag}"):
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_gather" in row["Kernel_Name"]:
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print(f"gran/loads/policy = {cfg}: " + ", ".join(f"{c}={sum(v.values()) / len(v):.0f}" for c, v in sorted(per.items())) + "  (mean per dispatch; 2^26 slots each)")
PY
  rm -rf $O/pmc_$tag $O/pmcb_$tag
done
cat $O/pmc_request_sizes.txt

#!/bin/bash
# Random-read ceiling of the HBM system for the access pattern of k_sites (tools/gather_ceiling.hip): rate by table size,
# request granularity and cache policy, then PMC passes that show which request sizes the L2 sends to the fabric.
# usage: tools/gather_sweep.sh OUTDIR        (run on the GPU box)
set -o pipefail
O=${1:-gpurun_out/gather}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $O
G=tools/_build/gather_ceiling
[ -x $G ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $G tools/gather_ceiling.hip || exit 1
: > $O/sweep.jsonl
for gib in 2 32 128; do
  for cfg in "128 2 0" "128 1 0" "128 2 1" "128 2 2" "128 2 3" "64 1 0" "64 1 1" "64 1 3" "32 1 0" "32 1 1" "32 1 3"; do
    timeout -k 5 120 $G $gib $cfg 28 32 >> $O/sweep.jsonl || { echo "gather $gib $cfg failed"; exit 1; }
  done
done
# occupancy: waves per CU (k_sites holds 32)
for w in 8 16 64; do timeout -k 5 120 $G 32 128 2 0 28 $w >> $O/sweep.jsonl || exit 1; done
cat $O/sweep.jsonl
rocprofv3 -L > $O/avail.txt 2>&1 || rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_RD_UNCACHED[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCC_EA0_RDREQ_DRAM[A-Za-z0-9_]*" $O/avail.txt | sort -u > $O/avail_rdreq.txt
cat $O/avail_rdreq.txt
# PMC: request sizes per form (32 GiB table)
for cfg in "128 2 0" "128 2 1" "128 2 3" "64 1 0" "64 1 1" "64 1 3" "32 1 0" "32 1 1" "32 1 3"; do
  tag=$(echo $cfg | tr ' ' _)
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $O/pmc_$tag -o p -- $G 32 $cfg 26 32 > $O/pmc_$tag.json 2> $O/pmc_$tag.log || { echo "pmc $cfg failed"; tail -3 $O/pmc_$tag.log; continue; }
  python3 - "$O/pmc_$tag" "$cfg" >> $O/pmc_request_sizes.txt <<'PY'
import csv, glob, sys, collections
d, cfg = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_gather" in row["Kernel_Name"]:
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print(f"gran/loads/policy = {cfg}: " + ", ".join(f"{c}={sum(v.values()) / len(v):.0f}" for c, v in sorted(per.items())) + "  (mean per dispatch of 2^26 slots)")
PY
  rm -rf $O/pmc_$tag
done
cat $O/pmc_request_sizes.txt

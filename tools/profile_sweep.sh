#!/bin/bash
# table-size sweep under PMC (GPU box): k_sites on configs[1] with ONE quad table of cores m = 13 .. 16 (2 GB .. 137 GB):
# lines fetched, L1-TLB misses, UTCL2 busy, read latency -- what the random gather costs as the table grows.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=${1:-gpurun_out/sweep}
mkdir -p $O
for m in 13 14 15 16; do
  export NEWMAP_AMD_QUAD_M=$m NEWMAP_AMD_QUAD_SMALL_M=0
  B="python3 bench.py --config c2 --no-cpu-baseline --no-end-to-end --no-spread --steps 10 --warmup 2"
  $B > $O/m$m.json 2> $O/m$m.log || echo "m=$m bench failed"
  timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/m$m/p1 -o p -- $B > /dev/null 2> $O/m$m.p1.log || echo "p1 failed"
  timeout -k 10 300 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY --output-format csv -d $O/m$m/p2 -o p -- $B > /dev/null 2> $O/m$m.p2.log || echo "p2 failed"
  timeout -k 10 300 rocprofv3 --pmc TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/m$m/p3 -o p -- $B > /dev/null 2> $O/m$m.p3.log || echo "p3 failed"
  python3 tools/pmc_summary.py k_sites $O/sweep_m${m}_summary.csv $O/m$m.json $O/m$m/p1 $O/m$m/p2 $O/m$m/p3 > /dev/null
  head -3 $O/sweep_m${m}_summary.csv
done

#!/bin/bash
# PMC passes over the headline of bench.py (3.09 Gbp, 20:200, one stream): HBM requests and TLB behaviour of
# k_sites<true, ...> per launch (mean over the 24 launches of a pass: 129 M positions each on average)
# usage: tools/profile_ns.sh OUTDIR
set -o pipefail
O=${1:-gpurun_out/prof_ns}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-configs1 --no-end-to-end --no-spread --streams 1 --steps 2 --warmup 1"
timeout -k 10 300 $B > $O/trace.json 2> $O/trace.log || { echo "plain run failed"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc1 -o p -- $B > $O/pmc1.json 2> $O/pmc1.log || { echo "pmc1 failed"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc2 -o p -- $B > $O/pmc2.json 2> $O/pmc2.log || echo "pmc2 failed"
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $O/pmc3 -o p -- $B > $O/pmc3.json 2> $O/pmc3.log || echo "pmc3 failed"
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc4 -o p -- $B > $O/pmc4.json 2> $O/pmc4.log || echo "pmc4 failed"
python3 tools/pmc_summary.py "k_sites<true" $O/pmc_ns_sites_kernel_summary.csv $O/trace.json $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4

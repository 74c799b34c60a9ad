#!/bin/bash
# PMC passes + kernel trace of one bench.py workload on one stream: per-dispatch means of the named kernels (HBM requests,
# TLB, SQ), as the MI355X guide prescribes (separate --pmc passes; no tracing in a counter pass).
# usage: tools/profile_cfg.sh CONFIG OUTDIR "KERNEL_SUBSTRING=NAME" ["KERNEL_SUBSTRING=NAME" ...] [-- extra bench flags]
#   e.g. tools/profile_cfg.sh hs gpurun_out/prof_hs "k_sweep<true=hs_k_sweep" "k_sites<true=hs_k_sites"
# writes OUTDIR/pmc_NAME_summary.csv per kernel and OUTDIR/CONFIG_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
set -o pipefail
CFG=$1; O=$2; shift 2
PAIRS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do PAIRS+=("$1"); shift; done; [ "$1" == "--" ] && shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $O
B="python3 bench.py --config $CFG --no-cpu-baseline --no-configs1 --no-end-to-end --no-spread --streams 1 --steps 2 --warmup 1 $*"
timeout -k 10 600 $B > $O/trace_$CFG.json 2> $O/trace_$CFG.log || { echo "plain run failed"; tail -5 $O/trace_$CFG.log; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$CFG -o k -- $B > /dev/null 2> $O/kt_$CFG.log || { echo "kernel trace failed"; exit 1; }
cp $(find $O/kt_$CFG -name '*kernel_stats.csv' | head -1) $O/${CFG}_kernel_stats_one_stream.csv && rm -rf $O/kt_$CFG
timeout -k 10 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc1 -o p -- $B > /dev/null 2> $O/pmc1.log || { echo "pmc1 failed"; exit 1; }
timeout -k 10 600 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc2 -o p -- $B > /dev/null 2> $O/pmc2.log || echo "pmc2 failed"
timeout -k 10 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $O/pmc3 -o p -- $B > /dev/null 2> $O/pmc3.log || echo "pmc3 failed"
timeout -k 10 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc4 -o p -- $B > /dev/null 2> $O/pmc4.log || echo "pmc4 failed"
for pr in "${PAIRS[@]}"; do
  python3 tools/pmc_summary.py "${pr%%=*}" $O/pmc_${pr##*=}_summary.csv $O/trace_$CFG.json $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 > /dev/null
  head -8 $O/pmc_${pr##*=}_summary.csv
done
rm -rf $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4

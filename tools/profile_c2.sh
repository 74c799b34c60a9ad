#!/bin/bash
# rocprofv3 evidence for the default bench (configs[1], headline only): kernel trace + PMC passes (run on the GPU box).
# Writes gpurun_out/prof_c2/{kernel_stats.csv, pmc_sites_kernel_summary.csv, ...}; copy the summaries to profiles/roundN/.
# Counters are collected in their own passes with --kernel-trace-free runs (the guide's HBM section).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=${1:-gpurun_out/prof_c2}
shift
mkdir -p $O
B="python3 bench.py --config c2 --no-cpu-baseline --no-end-to-end --no-spread --steps 20 --warmup 2 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace.json 2> $O/trace.log || echo "trace failed"
cp $(find $O/trace -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc1 -o p -- $B > $O/pmc1.json 2> $O/pmc1.log || echo "pmc1 failed"
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc2 -o p -- $B > $O/pmc2.json 2> $O/pmc2.log || echo "pmc2 failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/pmc3 -o p -- $B > $O/pmc3.json 2> $O/pmc3.log || echo "pmc3 failed"
timeout -k 10 300 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $O/pmc4 -o p -- $B > $O/pmc4.json 2> $O/pmc4.log || echo "pmc4 failed"
timeout -k 10 300 rocprofv3 --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_LFIFO_FULL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/pmc5 -o p -- $B > $O/pmc5.json 2> $O/pmc5.log || echo "pmc5 failed"
python3 tools/pmc_summary.py k_sites $O/pmc_sites_kernel_summary.csv $O/trace.json $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 $O/pmc5

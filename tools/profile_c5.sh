#!/bin/bash
# configs[4] (1 Gbp, 50 % tandem repeats, 20:255), 100 M launches: kernel trace + PMC passes over the probe kernels and k_resolve
# usage: tools/profile_c5.sh OUTDIR
set -o pipefail
O=${1:-gpurun_out/prof_c5}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p $O
B="python3 bench.py --config c5 --batch 100000000 --no-cpu-baseline --no-end-to-end --no-spread --streams 1 --steps 2 --warmup 1"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B > $O/trace.json 2> $O/trace.log || { echo "trace failed"; tail -5 $O/trace.log; exit 1; }
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv
rm -rf $O/trace
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc1 -o p -- $B > $O/pmc1.json 2> $O/pmc1.log || { echo "pmc1 failed"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -o p -- $B > $O/pmc2.json 2> $O/pmc2.log || echo "pmc2 failed"
for k in k_period_runs k_repeat_probe_coarse "k_repeat_probe<" k_resolve k_sites; do
  n=$(echo $k | tr -d '<')
  python3 tools/pmc_summary.py "$k" $O/pmc_c5_${n}_summary.csv $O/trace.json $O/pmc1 $O/pmc2 > /dev/null
done
rm -rf $O/pmc1 $O/pmc2
head -30 $O/pmc_c5_*_summary.csv
grep -h "k_sites\|k_resolve\|k_repeat" $O/c5_kernel_stats.csv | cut -d, -f1-6 | cut -c1-40,200-

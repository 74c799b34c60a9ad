#!/bin/bash
# rocprofv3 kernel trace of the north-star block of bench.py (3.09 Gbp, 20:200) and of configs[4]: per-kernel durations
# usage: tools/kernel_trace_ns.sh NAME
set -o pipefail
name=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$name
mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ns -o t -- python3 bench.py --no-cpu-baseline --no-configs1 --no-end-to-end --no-spread --streams 1 --steps 5 --warmup 1 > $O/bench_ns.json 2> $O/bench_ns.log || { echo "ns trace failed"; tail -5 $O/bench_ns.log; exit 1; }
f=$(find $O/trace_ns -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/ns_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5 -o t -- python3 bench.py --config c5 --batch 100000000 --no-cpu-baseline --no-end-to-end --no-spread --streams 1 --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.log || { echo "c5 trace failed"; tail -5 $O/bench_c5.log; exit 1; }
f=$(find $O/trace_c5 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/c5_kernel_stats.csv
rm -rf $O/trace_ns $O/trace_c5
grep -h "k_sites\|k_resolve\|k_repeat" $O/ns_kernel_stats.csv $O/c5_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,200-

#!/bin/bash
# rocprofv3 kernel trace of a bench.py run (GPU box): per-kernel durations as CSV under gpurun_out/<name>/
# usage: tools/kernel_trace.sh NAME [bench.py arguments ...]
set -o pipefail
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$name
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --no-configs1 --no-end-to-end --no-spread "$@" > $O/bench.json 2> $O/bench.log || { echo "trace failed"; tail -5 $O/bench.log; }
f=$(find $O/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats.csv && head -12 $O/kernel_stats.csv | cut -c1-200

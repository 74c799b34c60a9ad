#!/bin/bash
# kernel traces of the repeat-heavy and the full-size configs (run on the GPU box)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/prof_extra
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o t -- python3 bench.py --config c5 --mbp 200 --index-builder device --steps 5 --warmup 1 > $O/c5.json 2> $O/c5.log || echo "c5 trace failed"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o t -- python3 bench.py --config c3 --index-builder device --steps 3 --warmup 1 > $O/c3.json 2> $O/c3.log || echo "c3 trace failed"
head -8 $O/c5/t_kernel_stats.csv | cut -c1-160
head -8 $O/c3/t_kernel_stats.csv | cut -c1-160

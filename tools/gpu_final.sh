#!/bin/bash
# final trip of a round: PMC passes + kernel trace of the headline (tied to the sources as they are now), the full bench,
# the other BASELINE configs, the GPU suite
O=gpurun_out/${1:-final}
mkdir -p $O
bash tools/profile_c2.sh $O/prof_c2 > $O/prof_c2.txt 2>&1
mkdir -p profiles/round2 && cp $O/prof_c2/pmc_sites_kernel_summary.csv profiles/round2/   # bench.py below reads it
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --config c3 --steps 5 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python bench.py --config c5 --batch 100000000 --steps 5 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gputest.log
python tools/list_mode_timing.py --config c3 --passes 3 > $O/list_mode_c3.json 2> $O/list_mode_c3.err; echo "list mode rc=$?"

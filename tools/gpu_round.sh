#!/bin/bash
# one trip to the GPU box: traces at both launch sizes, the full bench, a capability config, PMC passes, the GPU tests
O=gpurun_out/${1:-round}
mkdir -p $O
bash tools/kernel_trace.sh $(basename $O)/trace_100m --steps 20 --warmup 3 --no-reference-batch > $O/trace_100m.txt 2>&1
bash tools/kernel_trace.sh $(basename $O)/trace_10m --steps 20 --warmup 3 --no-reference-batch --batch 10000000 > $O/trace_10m.txt 2>&1
bash tools/kernel_trace.sh $(basename $O)/trace_c5 --config c5 --steps 3 --warmup 1 --no-reference-batch > $O/trace_c5.txt 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --config c5 --steps 5 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
bash tools/profile_c2.sh $O/prof_c2 > $O/prof_c2.txt 2>&1
python -m pytest tests -m gpu -x -q --durations=8 > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/gputest.log

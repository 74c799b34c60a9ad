#!/bin/bash
# final trips of a round (run on the GPU box, one part per gpurun call: each stays below the call limit)
#   tools/gpu_final.sh b NAME   PMC passes tied to the sources as they are now (headline + configs[1]) -> profiles/round3/, then the
#                               default bench (reads them) and the --gpus 2 / 3 rehearsals from a bare shell
#   tools/gpu_final.sh c NAME   kernel traces (headline, configs[4]), PMC of configs[4]'s kernels, configs[2]
#   tools/gpu_final.sh d NAME   configs[4], list mode on the uniform and on the human-shaped 3 Gbp genome
#   tools/gpu_final.sh e NAME   the CLI process end to end, the A/B soak
#   tools/gpu_final.sh f NAME   configs[2] and the human-shaped genome, range mode
#   tools/gpu_final.sh a NAME   the GPU suite
part=$1
O=gpurun_out/${2:-final}
mkdir -p $O profiles/round3
case $part in
a)
  python -m pytest tests -m gpu -x -q --durations=12 > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gputest.log
  ;;
b)
  bash tools/profile_ns.sh $O/prof_ns > $O/prof_ns.txt 2>&1; tail -3 $O/prof_ns.txt
  cp $O/prof_ns/pmc_ns_sites_kernel_summary.csv profiles/round3/
  bash tools/profile_c2.sh $O/prof_c2 > $O/prof_c2.txt 2>&1
  cp $O/prof_c2/pmc_sites_kernel_summary.csv profiles/round3/; cp $O/prof_c2/kernel_stats.csv $O/c2_kernel_stats.csv
  python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/show_value.py $O/bench.json
  python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench (driver's flags) rc=$?"; python tools/show_value.py $O/bench_driver_flags.json
  NEWMAP_AMD_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 5 --warmup 1 --mbp 600 > $O/bench_n2_rehearsal.json 2> $O/bench_n2_rehearsal.err; echo "n2 rc=$?"
  NEWMAP_AMD_BENCH_REHEARSE=1 python bench.py --gpus 3 --steps 5 --warmup 1 --mbp 600 > $O/bench_n3_rehearsal.json 2> $O/bench_n3_rehearsal.err; echo "n3 rc=$?"
  NEWMAP_AMD_BENCH_REHEARSE=1 python bench.py --gpus 4 --steps 3 --warmup 1 --mbp 400 > $O/bench_n4_rehearsal.json 2> $O/bench_n4_rehearsal.err; echo "n4 rc=$?"
  ;;
c)
  bash tools/kernel_trace_ns.sh $(basename $O)/traces > $O/traces.txt 2>&1; tail -12 $O/traces.txt
  bash tools/profile_c5.sh $O/prof_c5 > $O/prof_c5.txt 2>&1; tail -3 $O/prof_c5.txt
  python bench.py --config c3 --steps 5 --warmup 2 --no-end-to-end > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; python tools/show_value.py $O/bench_c3.json
  ;;
d)
  # (no CPU baseline on the tandem genome: the oracle's comparison sort of whole suffixes is quadratic in a 50 kb array)
  python bench.py --config c5 --batch 100000000 --streams 3 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"; python tools/show_value.py $O/bench_c5.json
  python bench.py --config c5 --batch 100000000 --streams 5 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/bench_c5_s5.json 2> $O/bench_c5_s5.err; python tools/show_value.py $O/bench_c5_s5.json
  python tools/list_mode_timing.py --config c3 --passes 3 > $O/list_mode_c3.json 2> $O/list_mode_c3.err; echo "list mode c3 rc=$?"
  python tools/list_mode_timing.py --config hs --passes 3 > $O/list_mode_hs.json 2> $O/list_mode_hs.err; echo "list mode hs rc=$?"
  ;;
e)
  bash tools/kernel_trace_ns.sh $(basename $O)/traces > $O/traces.txt 2>&1; tail -6 $O/traces.txt
  bash tools/profile_c5.sh $O/prof_c5 > $O/prof_c5.txt 2>&1; tail -3 $O/prof_c5.txt
  python tools/e2e_timing.py --config c3 --device-index --out $O/e2e_c3.json > $O/e2e_c3.log 2>&1; echo "e2e rc=$?"
  python tools/fuzz_gpu.py --rounds 150 > $O/fuzz_gpu.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_gpu.log
  ;;
f)
  python bench.py --config c3 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; python tools/show_value.py $O/bench_c3.json
  python bench.py --config hs --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline --no-configs1 > $O/bench_hs.json 2> $O/bench_hs.err; echo "hs rc=$?"; python tools/show_value.py $O/bench_hs.json
  ;;
esac

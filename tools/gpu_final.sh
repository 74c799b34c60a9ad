#!/bin/bash
# final trips of round 4 (run on the GPU box, one part per gpurun call: each stays below the call limit)
#   tools/gpu_final.sh a NAME   the GPU suite
#   tools/gpu_final.sh b NAME   PMC passes tied to the sources as they are now (headline k_sites + its one- and two-stream kernel traces,
#                               configs[1]) -> profiles/round4/
#   tools/gpu_final.sh c NAME   PMC passes of the human-shaped genome (k_sweep, k_sites) and of configs[4]'s kernels -> profiles/round4/
#   tools/gpu_final.sh d NAME   the default bench (reads the summaries), the driver's flags, the --gpus 2 rehearsal
#   tools/gpu_final.sh e NAME   configs[2], configs[4] (3 and 5 streams), the human-shaped genome, list mode on both 3 Gbp genomes
#   tools/gpu_final.sh f NAME   the CLI process end to end, the A/B soak
part=$1
O=gpurun_out/${2:-final}
P=$O/profiles_round4        # (only gpurun_out/ comes back from the box: copy from here into profiles/round4/ afterwards)
mkdir -p $O $P
case $part in
a)
  python -m pytest tests -m gpu -x -q --durations=15 > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -22 $O/gputest.log
  ;;
b)
  bash tools/profile_cfg.sh ns $O/prof_ns "k_sites<true=ns_k_sites" > $O/prof_ns.txt 2>&1; tail -3 $O/prof_ns.txt
  cp $O/prof_ns/pmc_ns_k_sites_summary.csv $O/prof_ns/ns_kernel_stats_one_stream.csv $P/
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -o k -- python3 bench.py --no-cpu-baseline --no-configs1 --no-end-to-end --no-spread --steps 5 --warmup 1 > $O/bench_two_streams.json 2> $O/kt2.log || echo "two-stream trace failed"
  cp $(find $O/kt2 -name '*kernel_stats.csv' | head -1) $P/ns_kernel_stats_two_streams.csv && rm -rf $O/kt2
  grep -h "k_sites" $P/ns_kernel_stats_one_stream.csv $P/ns_kernel_stats_two_streams.csv | cut -d, -f1-4 | cut -c1-40,330-
  bash tools/profile_cfg.sh c2 $O/prof_c2 "k_sites<false=c2_k_sites" > $O/prof_c2.txt 2>&1; tail -3 $O/prof_c2.txt
  cp $O/prof_c2/pmc_c2_k_sites_summary.csv $O/prof_c2/c2_kernel_stats_one_stream.csv $P/
  ;;
c)
  bash tools/profile_cfg.sh hs $O/prof_hs "k_sweep<true=hs_k_sweep" "k_sites<true=hs_k_sites" > $O/prof_hs.txt 2>&1; tail -3 $O/prof_hs.txt
  cp $O/prof_hs/pmc_hs_k_sweep_summary.csv $O/prof_hs/pmc_hs_k_sites_summary.csv $O/prof_hs/hs_kernel_stats_one_stream.csv $P/
  bash tools/profile_cfg.sh c5 $O/prof_c5 "k_sites<false=c5_k_sites" "k_period_runs<false=c5_k_period_runs" "k_repeat_probe<false=c5_k_repeat_probe" "k_resolve<false=c5_k_resolve" -- --batch 100000000 > $O/prof_c5.txt 2>&1; tail -3 $O/prof_c5.txt
  cp $O/prof_c5/pmc_c5_*_summary.csv $O/prof_c5/c5_kernel_stats_one_stream.csv $P/
  ;;
d)
  python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/show_value.py $O/bench.json
  python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err; echo "bench (driver's flags) rc=$?"; python tools/show_value.py $O/bench_driver_flags.json
  NEWMAP_AMD_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 5 --warmup 1 --mbp 600 > $O/bench_n2_rehearsal.json 2> $O/bench_n2_rehearsal.err; echo "n2 rc=$?"
  ;;
e)
  python bench.py --config c3 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"; python tools/show_value.py $O/bench_c3.json
  python bench.py --config c5 --batch 100000000 --streams 3 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/bench_c5_s3.json 2> $O/bench_c5_s3.err; echo "c5 rc=$?"; python tools/show_value.py $O/bench_c5_s3.json
  python bench.py --config c5 --batch 100000000 --streams 5 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/bench_c5_s5.json 2> $O/bench_c5_s5.err; python tools/show_value.py $O/bench_c5_s5.json
  python bench.py --config hs --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/bench_hs.json 2> $O/bench_hs.err; echo "hs rc=$?"; python tools/show_value.py $O/bench_hs.json
  python tools/list_mode_timing.py --config hs --passes 3 > $O/list_mode_hs.json 2> $O/list_mode_hs.err; echo "list mode hs rc=$?"
  python tools/list_mode_timing.py --config c3 --passes 3 > $O/list_mode_c3.json 2> $O/list_mode_c3.err; echo "list mode c3 rc=$?"
  ;;
f)
  python tools/e2e_timing.py --config c3 --device-index --out $O/e2e_c3.json > $O/e2e_c3.log 2>&1; echo "e2e rc=$?"
  python tools/fuzz_gpu.py --rounds 150 > $O/fuzz_gpu.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_gpu.log
  ;;
esac

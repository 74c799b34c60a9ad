O=gpurun_out/r2o
mkdir -p $O
python tools/list_mode_timing.py --config c3 --passes 3 > $O/list_mode_c3.json 2> $O/list_mode_c3.err; echo "list mode rc=$?"
python tools/list_mode_timing.py --config c2 --passes 5 > $O/list_mode_c2.json 2> $O/list_mode_c2.err; echo "list mode c2 rc=$?"
python -m pytest tests -m gpu -x -q -k "config3_full" > $O/gputest_c3.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputest_c3.log
python tools/fuzz_gpu.py --rounds 100 --seed 5 > $O/fuzz_gpu_100_rounds.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_gpu_100_rounds.log

#!/bin/bash
# second kind of trip: the repeat-rich config with the probes beside / behind k_sites, a 2-rank rehearsal of the N > 1 bench
# code path on the one GPU (gloo), the table-size sweep, selected GPU tests
O=gpurun_out/${1:-round}
mkdir -p $O
python bench.py --config c5 --steps 5 --warmup 2 > $O/bench_c5_beside.json 2> $O/bench_c5_beside.err; echo "c5 beside rc=$?"
NEWMAP_AMD_PROBES_BESIDE=0 python bench.py --config c5 --steps 5 --warmup 2 > $O/bench_c5_behind.json 2> $O/bench_c5_behind.err; echo "c5 behind rc=$?"
NEWMAP_AMD_BENCH_REHEARSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --seed-length auto-small --north-star-mbp 400 --no-cpu-baseline > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc=$?"; tail -3 $O/bench_n2.err
bash tools/profile_sweep.sh $O/sweep > $O/sweep.txt 2>&1
python -m pytest tests -m gpu -x -q -k "kernels_agree or big_index or coarse or soak or repeat_probes or config5 or front_ends" > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputest.log

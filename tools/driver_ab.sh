#!/bin/bash
# native driver on the 3.09 Gbp genome by number of streams (copies in flight) and workers; NEWMAP_AMD_DRIVER_ZEROCOPY=1: the
# kernels read the pinned slot over the bus themselves instead of a DMA copy-in
O=gpurun_out/${1:-drv_ab}
mkdir -p $O
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs1 --no-end-to-end --no-spread > $O/bench.json 2> $O/bench.err
for st in 3 4 5; do
  NEWMAP_AMD_DRIVER_STREAMS=$st python tools/driver_sweep.py --workers 8 10 12 16 > $O/sweep_st${st}.jsonl 2> $O/sweep_st${st}.err
  echo "streams $st"; cat $O/sweep_st${st}.jsonl; grep "\[driver\]" $O/sweep_st${st}.err | awk 'NR%3==0'
done

#!/bin/bash
# native driver, 3.09 Gbp FASTA -> 24 files: with and without the sweep's launches (same box)
O=gpurun_out/r4e; mkdir -p $O
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs1 --no-end-to-end --no-spread > $O/bench.json 2> $O/bench.err
for sw in 1 0 1 0; do
  NEWMAP_AMD_SWEEP=$sw python tools/driver_sweep.py --workers 10 > $O/drv_sweep$sw.jsonl 2> $O/drv_sweep$sw.err
  echo "sweep=$sw $(cat $O/drv_sweep$sw.jsonl)"; grep "\[driver\]" $O/drv_sweep$sw.err | tail -1
done

#!/bin/bash
# configs[2] (3.09 Gbp, 24:150) with and without the repeat dictionary / the two-base LF blocks; run on the GPU box
O=gpurun_out/${1:-ab_c3}
mkdir -p $O
one() { python bench.py --config c3 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/c3_$1.json 2> $O/c3_$1.err; echo "c3 $1 rc=$?"; python tools/show_value.py $O/c3_$1.json; }
one default
NEWMAP_AMD_DICT=0 one nodict
NEWMAP_AMD_LF2=0 one nolf2
NEWMAP_AMD_DICT=0 NEWMAP_AMD_LF2=0 one neither

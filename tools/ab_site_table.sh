#!/bin/bash
# which quad table the sites of the headline read: picked per launch (default: cores of 14, second chance on 15), the long
# cores (NEWMAP_AMD_SITE_TABLE=1: 6 positions per line, 2.3 % open, no second chance: the dictionary takes them), the short ones
O=gpurun_out/${1:-ab_site_table}
mkdir -p $O
one() { python bench.py --config $1 --steps 8 --warmup 2 --no-end-to-end --no-cpu-baseline --no-configs1 > $O/$1_$2.json 2> $O/$1_$2.err; echo "$1 $2 rc=$?"; python tools/show_value.py $O/$1_$2.json; }
one ns default
NEWMAP_AMD_SITE_TABLE=1 one ns long_cores
one c3 default
NEWMAP_AMD_SITE_TABLE=1 one c3 long_cores

#!/bin/bash
# round 4: number of streams with eight hardware queues -- headline, configs[1]'s reference batch, hs
O=gpurun_out/r4h; mkdir -p $O
one() { cfg=$1; name=$2; shift 2; python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end --no-configs1 --no-spread "$@" > $O/$name.json 2> $O/$name.log; python tools/show_value.py $O/$name.json; }
one ns ns_s2
one ns ns_s3 --streams 3
one ns ns_s4 --streams 4
one ns ns_s5 --streams 5
one hs hs_s5 --streams 5 --steps 10
one c3 c3_s2 --steps 10
one c3 c3_s4 --streams 4 --steps 10

#!/bin/bash
# A/B of the two-base LF blocks on the workloads where walks are long (configs[4], the human-shaped stand-in) and where they
# are short (the headline); run on the GPU box.   tools/ab_lf2.sh NAME [wide ...]
#   NEWMAP_AMD_LF2=0|1 : blocks built or not;  NM_LF2_WIDE (compile time): rows from which a walk tries two bases whatever its history
O=gpurun_out/${1:-ab_lf2}
shift
mkdir -p $O
one() {   # tag
  NEWMAP_AMD_VERBOSE=1 python bench.py --config c5 --batch 100000000 --streams 3 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/c5_$1.json 2> $O/c5_$1.err; echo "c5 $1 rc=$?"; python tools/show_value.py $O/c5_$1.json
  python bench.py --config hs --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline --no-configs1 > $O/hs_$1.json 2> $O/hs_$1.err; echo "hs $1 rc=$?"; python tools/show_value.py $O/hs_$1.json
  NEWMAP_AMD_VERBOSE=1 python bench.py --steps 10 --warmup 3 --no-end-to-end --no-cpu-baseline > $O/ns_$1.json 2> $O/ns_$1.err; echo "ns $1 rc=$?"; python tools/show_value.py $O/ns_$1.json
}
[ -n "$SKIP_OFF" ] || NEWMAP_AMD_LF2=0 one off
for wide in ${@:-32}; do
  touch newmap_amd/csrc/nm_engine.hip
  make -C newmap_amd/csrc HIPFLAGS_EXTRA=-DNM_LF2_WIDE=${wide}u > $O/make_$wide.log 2>&1 || { echo "make failed"; exit 1; }
  one wide$wide
done

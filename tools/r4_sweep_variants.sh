#!/bin/bash
# round 4, last calls: variants of k_sweep against the product library on one box (human-shaped 3.09 Gbp genome, 20:200).
# The variants are compile-time switches that are NOT in the tree: apply profiles/round4/sweep_variants.patch first
# (NM_SWEEP_CHUNK / NM_SWEEP_MAX_BLOCKS overridable, NM_SWEEP_CLOCK, NM_SWEEP_GUIDED), then
#   tools/r4_sweep_variants.sh build NAME "FLAGS"      here, e.g.  build M2 "-DNM_SWEEP_MAX_BLOCKS=512u"
#   tools/r4_sweep_variants.sh run OUT NAME:STREAMS ... on the GPU box, e.g.  run r4_occ4 M2:5 base:5
# Results of round 4: profiles/round4/ab_sweep_variants.json.
set -o pipefail
D=tools/_build
case $1 in
build)
  mkdir -p $D
  ( cd newmap_amd/csrc && make -s && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $3 -c -o /tmp/nm_engine_$2.o nm_engine.hip \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../$D/libnewmap_amd_$2.so nm_build.o /tmp/nm_engine_$2.o nm_driver.o nm_track.o nm_build_device.o -lz -lgomp ) && echo "built $D/libnewmap_amd_$2.so"
  ;;
run)
  O=gpurun_out/$2; mkdir -p $O; shift 2
  for v in "$@"; do
    name=${v%%:*}; streams=${v#*:}
    lib=$PWD/$D/libnewmap_amd_$name.so; [ $name = base ] && lib=$PWD/newmap_amd/libnewmap_amd.so
    NEWMAP_AMD_RAW_TALLIES=1 NEWMAP_AMD_LIB=$lib timeout -k 10 170 python bench.py --config hs --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --no-configs1 --no-spread --streams $streams \
      > $O/${name}_s$streams.json 2> $O/${name}_s$streams.log || { echo "$name failed"; exit 1; }
    python tools/show_value.py $O/${name}_s$streams.json
  done
  ;;
esac

#!/bin/bash
# round 4: the sweep with the index's LCP bytes -- parity subset, soak, hs with and without them (same box)
set -o pipefail
O=gpurun_out/r4f; mkdir -p $O
run() { # name, env..., -- bench args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 600 python bench.py "$@" --no-cpu-baseline --no-end-to-end --no-configs1 --no-spread > $O/$name.json 2> $O/$name.log || { echo "$name failed"; tail -5 $O/$name.log; return 1; }
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']
print(sys.argv[2], round(d['value']/1e9,2), 'G/s', {n[:12]:round(v['total_ms'],1) for n,v in k.items()}, d['pipeline']['resolve'], 'open_s', round(d['host']['index_open_s'],1), 'build_s', round(d['host']['index_build_s'],1), flush=True)
PY
}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fixtures or human_shaped_stand_in or device_index_builder or device_built_index or config5_tandem or min_unique_equals_oracle" > $O/pytest_subset.log 2>&1 || { tail -30 $O/pytest_subset.log; exit 1; }
tail -1 $O/pytest_subset.log
timeout -k 10 600 python tools/fuzz_gpu.py --rounds 30 --seed 61 > $O/fuzz.log 2>&1 || { tail -30 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
run hs_lcp X=1 -- --config hs || exit 1
#run hs_nolcp NEWMAP_AMD_LCP=0 -- --config hs
#run c5_lcp X=1 -- --config c5 --batch 100000000 --streams 3

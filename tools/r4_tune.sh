#!/bin/bash
# round 4: routing between k_sweep and k_resolve on one box: soak, hs, c5, headline, native driver with and without the sweep's launches
set -o pipefail
O=gpurun_out/r4b; mkdir -p $O
run() { # name, env..., -- bench args
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 600 python bench.py "$@" --no-cpu-baseline --no-end-to-end --no-configs1 --no-spread > $O/$name.json 2> $O/$name.log || { echo "$name failed"; tail -5 $O/$name.log; return 1; }
  python - $O/$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']
print(sys.argv[2], round(d['value']/1e9,2), 'G/s', {n[:12]:round(v['total_ms'],1) for n,v in k.items()}, d['pipeline']['resolve'].get('sweep'), flush=True)
PY
}
timeout -k 10 600 python tools/fuzz_gpu.py --rounds 25 --seed 51 > $O/fuzz.log 2>&1 || { tail -30 $O/fuzz.log; exit 1; }
tail -1 $O/fuzz.log
run hs_default X=1 -- --config hs || exit 1
run c5_default X=1 -- --config c5 --batch 100000000 --streams 3
run ns_default X=1 -- --config ns
for sw in 1 0 1 0; do
  NEWMAP_AMD_SWEEP=$sw python tools/driver_sweep.py --workers 10 > $O/drv_sweep$sw.jsonl 2> $O/drv_sweep$sw.err
  echo "sweep=$sw $(cat $O/drv_sweep$sw.jsonl)"; grep "\[driver\]" $O/drv_sweep$sw.err | tail -1
done

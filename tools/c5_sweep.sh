#!/bin/bash
# A/B sweep on the tandem-repeat config at 200 Mbp (measurement helper, not part of the product)
mkdir -p gpurun_out/sweep5
run() {
    name=$1; shift
    env "$@" timeout -k 10 300 python bench.py --config c5 --mbp 200 --index-builder device --steps 5 --warmup 1 > gpurun_out/sweep5/$name.json 2> gpurun_out/sweep5/$name.log || echo "$name failed"
    python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/sweep5/{name}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{name:24s} {d['value']/1e9:7.2f} G/s  step {d['ms_per_step']:.3f} ms  {r['kernel']} {r['avg_launch_ms']*1e3:7.1f} us  lf/pos {r['lf_steps_per_position']:.3f}  probe lf/pos {d['repeat_probes']['lf_steps_per_position']:.3f}  verify {d.get('verify',{}).get('sampled_positions')}", flush=True)
except Exception as e:
    print(name, "no result", e, flush=True)
PY
}
for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    run $name $(echo $envs | tr ',' ' ')
done

#!/bin/bash
# A/B sweep of the range kernels on configs[1] (measurement helper, not part of the product)
mkdir -p gpurun_out/sweep
run() {  # name, env...
    name=$1; shift
    env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/sweep/$name.json 2> gpurun_out/sweep/$name.log || echo "$name failed"
    python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/sweep/{name}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{name:28s} {d['value']/1e9:7.2f} G/s  step {d['ms_per_step']:.3f} ms  kernel {r['kernel']} {r['avg_launch_ms']*1e3:7.1f} us  lf/pos {r['lf_steps_per_position']:.4f}  open {d['host']['index_open_s']:.2f}s  hbm {d['config']['index_bytes_hbm']/1e9:.0f} GB", flush=True)
except Exception as e:
    print(name, "no result", e, flush=True)
PY
}
for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    run $name $(echo $envs | tr ',' ' ')
done

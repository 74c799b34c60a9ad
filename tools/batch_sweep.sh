#!/bin/bash
# configs[1] with other --kmer-batch-size values (measurement helper)
mkdir -p gpurun_out/sweepb
for b in "$@"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --batch $b > gpurun_out/sweepb/b$b.json 2> gpurun_out/sweepb/b$b.log || echo "$b failed"
    python - $b <<'PY'
import json, sys
b = sys.argv[1]
d = json.loads(open(f"gpurun_out/sweepb/b{b}.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(f"batch {int(b)/1e6:6.1f} M  {d['value']/1e9:7.2f} G/s  step {d['ms_per_step']:.3f} ms  kernel {r['avg_launch_ms']*1e3:7.1f} us x {d['config']['segments_per_rank']}", flush=True)
PY
done

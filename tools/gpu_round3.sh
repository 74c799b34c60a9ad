#!/bin/bash
# c5 with coarse strides 128 / 256 / 512
O=gpurun_out/${1:-round}
mkdir -p $O
for cs in 512 256 128; do
  NEWMAP_AMD_COARSE_STRIDE=$cs python bench.py --config c5 --steps 5 --warmup 2 --no-reference-batch > $O/bench_c5_cs$cs.json 2> $O/bench_c5_cs$cs.err; echo "c5 cs=$cs rc=$?"
  python - $O/bench_c5_cs$cs.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  value", round(r["value"] / 1e9, 1), "ms/step", round(r["ms_per_step"], 3), "pipe", round(r["pipeline"]["avg_segment_ms"], 3), r["pipeline"]["repeat_probes"], flush=True)
PY
done
NEWMAP_AMD_COARSE_STRIDE=128 bash tools/kernel_trace.sh $(basename $O)/trace_c5_128 --config c5 --steps 3 --warmup 1 --no-reference-batch > $O/trace_c5_128.txt 2>&1

#!/bin/bash
# bench.py on configs[1] for several quad-table core lengths (NEWMAP_AMD_QUAD_M); one JSON line per run
out=${1:-gpurun_out/sweep_m}
mkdir -p "$out"
for m in 16 15 14 13 12; do
  NEWMAP_AMD_QUAD_M=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$out/m$m.json" 2> "$out/m$m.err" || { echo "m=$m failed"; tail -3 "$out/m$m.err"; }
  python - "$out/m$m.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r["config"]["quad_core_length"], "ms/step", round(r["ms_per_step"], 4), "kernel ms", round(r["roofline"]["avg_launch_ms"], 4),
      "Gpos/s", round(r["value"] / 1e9, 1), "ref-batch", round(r["reference_batch"]["value"] / 1e9, 1),
      "words/pos", round(r["roofline"]["table_words_per_position"], 4), "lf/pos", round(r["roofline"]["lf_steps_per_position"], 5),
      "bytes", r["config"]["index_bytes_hbm"] >> 30, "GiB", flush=True)
PY
done

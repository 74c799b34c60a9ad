#!/bin/bash
O=gpurun_out/${1:-round}
mkdir -p $O
bash tools/cli_ranks_rehearsal.sh > $O/cli_ranks.txt 2>&1; echo "cli ranks rc=$?"; tail -4 $O/cli_ranks.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-reference-batch > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n = r["north_star"]
print("c2", round(r["value"] / 1e9, 1), "north star", round(n["value"] / 1e9, 1), "ms/step", round(n["ms_per_step"], 2), "launches", n["launches_per_step_per_rank"], "e2e", r["end_to_end"]["cli_search_s"])
PY

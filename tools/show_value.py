"""print value (G positions/s) and ms_per_step of bench.py output files"""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e9, 2), "G pos/s", round(d["ms_per_step"], 3), "ms", flush=True)

#!/bin/bash
# the repeat dictionary after the second chance (default) / instead of it (NEWMAP_AMD_SEED_POLICY=0x2000) / not at all
# (NEWMAP_AMD_DICT=0): headline (with its configs[1] block), configs[2], configs[4], the human-shaped genome
O=gpurun_out/${1:-ab_dict}
mkdir -p $O
one() { python bench.py --config $1 --steps 8 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/$1_$2.json 2> $O/$1_$2.err; echo "$1 $2 rc=$?"; python tools/show_value.py $O/$1_$2.json; }
for c in ns c3; do
  one $c default
  NEWMAP_AMD_SEED_POLICY=0x2000 one $c dict_alone
  NEWMAP_AMD_DICT=0 one $c nodict
done
python bench.py --config c5 --batch 100000000 --streams 3 --steps 5 --warmup 2 --no-end-to-end --no-cpu-baseline > $O/c5_default.json 2> $O/c5_default.err; python tools/show_value.py $O/c5_default.json
python bench.py --config hs --steps 3 --warmup 1 --no-end-to-end --no-cpu-baseline --no-configs1 > $O/hs_default.json 2> $O/hs_default.err; python tools/show_value.py $O/hs_default.json
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
for t in ("default", "dict_alone", "nodict"):
    d = json.loads(open(f"{o}/ns_{t}.json").read().strip().splitlines()[-1])
    c = d.get("configs1")
    if c: print("configs1", t, round(c["value"] / 1e9, 1), "G", c.get("ms_per_step"))
PY

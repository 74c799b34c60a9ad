#!/bin/bash
# `newmap search` as a 2-rank torch.distributed job with both ranks on the one GPU (gloo barriers), against the
# single-process run: the files must be identical.  GPU box only.
set -e
W=$(mktemp -d /tmp/newmap_cli_XXXX)
python - "$W" <<'PY'
import sys, numpy as np
from pathlib import Path
sys.path.insert(0, ".")
from newmap_amd import synth
w = Path(sys.argv[1])
recs = [("chrA", synth.uniform_dna(30_000_000, 5)), ("chrB", synth.uniform_dna(7_000_001, 6)), ("chrC", synth.tandem_dna(3_000_000, 7))]
recs[1][1][1_000_000:1_000_500] = ord("N")
synth.write_fasta(w / "g.fa", recs)
PY
cd "$W"
export PYTHONPATH=$GRAFT_REPO_ROOT
python -m newmap_amd.main index g.fa --device 0 > /dev/null
mkdir one two
python -m newmap_amd.main search g.fa --search-range 20:200 -o one --device 0
NEWMAP_AMD_DIST_BACKEND=gloo NEWMAP_AMD_SHARD_CHUNK=3000000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 \
    -m newmap_amd.main search g.fa --search-range 20:200 -o two --device 0
for f in one/*; do cmp "$f" "two/$(basename $f)"; done
ls -la one two | head -12
echo "CLI ranks rehearsal: files identical"
rm -rf "$W"

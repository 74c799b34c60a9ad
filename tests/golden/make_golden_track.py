#!/usr/bin/env python3
"""tests/golden/golden_track.json: outputs of the REFERENCE's newmap/track.py
(write_mappability_files) on small unique-length arrays.  Build container only (/root/reference)."""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

sys.path.insert(0, "/root/reference")
from newmap import track  # noqa: E402

HERE = Path(__file__).resolve().parent
rng = np.random.default_rng(20260607)
arrays = {
    "chr1": ([0, 10, 9, 8, 7, 6, 5, 4, 4, 4, 6, 5, 4, 4, 4, 0, 0, 0, 0, 0], "uint8"),
    "chr2": ([10, 10, 9, 8, 7, 6, 5, 4, 4, 4] + [0] * 20, "uint8"),
    "rnd": (np.where(rng.random(400) < 0.3, 0, rng.integers(4, 40, 400)).tolist(), "uint8"),
    "wide.name": (np.where(rng.random(300) < 0.5, 0, rng.integers(20, 400, 300)).tolist(), "uint16"),
    "allzero": ([0] * 25, "uint8"),
}
cases = []
with tempfile.TemporaryDirectory() as td:
    td = Path(td)
    for name, (vals, dt) in arrays.items():
        np.array(vals, dtype=dt).tofile(td / f"{name}.unique.{dt}")
    for k in (1, 5, 10, 24, 100, 150):
        files = [td / f"{n}.unique.{dt}" for n, (_, dt) in arrays.items()]
        bed, wig = td / "o.bed", td / "o.wig"
        track.write_mappability_files(files, k, str(bed), str(wig), False)
        cases.append({"k": k, "bed": bed.read_text(), "wig": wig.read_text()})
(HERE / "golden_track.json").write_text(json.dumps(
    {"arrays": {n: {"values": v, "dtype": dt} for n, (v, dt) in arrays.items()}, "cases": cases}))
print("wrote", len(cases), "track cases")

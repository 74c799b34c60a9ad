#!/usr/bin/env python3
"""tests/golden/golden_multi.json: the REFERENCE driver run with several FASTA files in lock-step and
several index files (newmap/search.py:251-265, 461, 656-697 -- the bisulfite-style mode).  Build
container only (/root/reference); the absent native counter is replaced at the FFI seam exactly as in
make_golden.py."""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as mg  # noqa: E402


def run_multi(search, fastas, lengths, is_binary, batch, **kw):
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        fa_paths, idx_paths = [], []
        for i, text in enumerate(fastas):
            fa = td / f"in{i}.fa"
            fa.write_bytes(text)
            fa_paths.append(fa)
            idx = str(td / f"in{i}.awfmi")
            mg.register_index(idx, text)
            idx_paths.append(Path(idx))
        out = td / "out"
        out.mkdir()
        cfg = search.SearchConfig(fasta_filepaths=fa_paths, fmindex_filepaths=idx_paths, kmer_lengths=list(lengths),
                                  is_binary_search=is_binary, kmer_batch_size=batch, output_directory=out, **kw)
        search.write_unique_counts(cfg)
        res = {}
        for f in sorted(out.iterdir()):
            suffix = f.name.rsplit(".", 1)[1]
            res[f.name.rsplit(".", 2)[0]] = {"dtype": suffix, "values": np.fromfile(f, dtype=suffix).tolist()}
        return res


def main():
    fasta, search = mg.import_reference()
    rng = np.random.default_rng(20260608)
    base = bytearray(mg.random_dna(rng, 2500))
    base[700:900] = base[100:300]                      # a repeat
    base[1500:1540] = b"N" * 40
    g = bytes(base)
    ct = g.replace(b"C", b"T")                          # bisulfite-style conversions keep N positions equal
    ga = g.replace(b"G", b"A")
    second = mg.random_dna(rng, 900)
    f_ct = mg.fasta_text([("chrA", ct), ("chrB", second.replace(b"C", b"T"))])
    f_ga = mg.fasta_text([("chrA", ga), ("chrB", second.replace(b"G", b"A"))])
    cases = []

    def add(name, lengths, is_binary, batch, **kw):
        cases.append({"name": name, "fastas": [f_ct.decode("latin-1"), f_ga.decode("latin-1")],
                      "kmer_lengths": list(lengths), "is_binary": is_binary, "batch": batch,
                      "use_reverse_complement": kw.get("use_reverse_complement", True),
                      "expected": run_multi(search, [f_ct, f_ga], lengths, is_binary, batch, **kw)})

    add("M1_two_fasta_two_index_10_80", range(10, 81), True, 10_000_000)
    add("M1_batches_of_600", range(10, 81), True, 600)
    add("M2_norc", range(10, 81), True, 10_000_000, use_reverse_complement=False)
    add("M3_list_20_40", [20, 40], False, 10_000_000)
    (HERE / "golden_multi.json").write_text(json.dumps({"cases": cases}))
    for c in cases:
        print(c["name"], {k: int(np.count_nonzero(v["values"])) for k, v in c["expected"].items()})


if __name__ == "__main__":
    main()

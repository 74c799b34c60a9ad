#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REFERENCE's own Python driver.

Runs only in the build container (needs /root/reference); the fixtures it writes are data --
inputs and the reference's outputs -- and are what travels to the GPU box.

The reference's native counter (AvxWindowFmIndex, absent from /root/reference) is replaced at the
FFI seam ``newmap._c_newmap_count_kmers.count_kmers_from_sequence`` (newmap/search.py:12) by a
pure-Python brute-force counter written here, independent of oracle/: occurrences of the
k-mer inside single FASTA records of the indexed file, case-insensitive, with every non-ACGT byte
one equivalent letter (docs/source/commands.rst:134-144).  It reproduces all four known-answer
vectors of the reference's tests (asserted below), which is the evidence that the seam model is
the right one.

Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import io
import json
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent

# ---------------------------------------------------------------- stand-in at the FFI seam
_FOLD = bytearray(b"X" * 256)
for a in b"ACGT":
    _FOLD[a] = a
    _FOLD[a + 32] = a
_FOLD = bytes(_FOLD)

_INDEXES: dict[str, list[bytes]] = {}      # "index path" -> folded records


def _parse_records(text: bytes) -> list[bytes]:
    recs, cur = [], []
    for line in text.splitlines():
        line = line.rstrip()
        if line.startswith((b">", b";")):
            if cur:
                recs.append(b"".join(cur))
            cur = []
        else:
            cur.append(line)
    if cur:
        recs.append(b"".join(cur))
    return [r for r in recs if r]


def register_index(path: str, fasta_text: bytes):
    _INDEXES[path] = [r.translate(_FOLD) for r in _parse_records(fasta_text)]


def _count(records: list[bytes], kmer: bytes) -> int:
    k = kmer.translate(_FOLD)
    total = 0
    for r in records:
        i = r.find(k)
        while i >= 0:
            total += 1
            i = r.find(k, i + 1)
    return total


def count_kmers(index_path, kmers, num_threads):
    recs = _INDEXES[index_path]
    return [_count(recs, k) for k in kmers]


def count_kmers_from_sequence(index_path, sequence, starts, lengths, num_threads):
    recs = _INDEXES[index_path]
    cache: dict[bytes, int] = {}
    out = []
    for s, l in zip(starts, lengths):
        k = bytes(sequence[s:s + l])
        if k not in cache:
            cache[k] = _count(recs, k)
        out.append(cache[k])
    return out


def import_reference():
    sys.path.insert(0, str(REF))
    stub = types.ModuleType("newmap._c_newmap_count_kmers")
    stub.count_kmers = count_kmers
    stub.count_kmers_from_sequence = count_kmers_from_sequence
    sys.modules["newmap._c_newmap_count_kmers"] = stub
    import newmap.fasta as fasta          # noqa: E402
    import newmap.search as search        # noqa: E402
    return fasta, search


# ---------------------------------------------------------------- inputs
def fasta_text(records: list[tuple[str, bytes]], width: int = 60) -> bytes:
    out = []
    for name, seq in records:
        out.append(b">" + name.encode())
        out.extend(seq[i:i + width] for i in range(0, len(seq), width))
    return b"\n".join(out) + b"\n"


def random_dna(rng, n):
    return bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)])


def tandem_dna(rng, n, frac=0.5):
    """alternate unique spacers and tandem arrays until n bases, ~frac of them in arrays"""
    out = bytearray()
    while len(out) < n:
        out += random_dna(rng, int(rng.integers(20, 200)))
        unit = random_dna(rng, int(rng.integers(2, 40)))
        copies = int(rng.integers(3, 30))
        if rng.random() < frac * 2:
            out += unit * copies
    return bytes(out[:n])


def run_search(search, fasta_bytes: bytes, lengths, is_binary, batch, **kw):
    """Run the reference's write_unique_counts and collect {id: list} of its output files."""
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        fa = td / "in.fa"
        fa.write_bytes(fasta_bytes)
        idx = str(td / "in.awfmi")
        register_index(idx, fasta_bytes)
        outdir = td / "out"
        outdir.mkdir()
        cfg = search.SearchConfig(fasta_filepaths=[fa], fmindex_filepaths=[Path(idx)],
                                  kmer_lengths=list(lengths), is_binary_search=is_binary,
                                  kmer_batch_size=batch, output_directory=outdir, **kw)
        search.write_unique_counts(cfg)
        res = {}
        for f in sorted(outdir.iterdir()):
            rid, _, suffix = f.name.rsplit(".", 2)[0], None, f.name.rsplit(".", 1)[1]
            res[rid] = {"dtype": suffix, "values": np.fromfile(f, dtype=suffix).tolist()}
        return res


def main():
    fasta, search = import_reference()
    genome = (REF / "tests/data/genome.fa").read_bytes()
    chr2 = (REF / "tests/data/chr2.fa").read_bytes()
    (HERE / "genome.fa").write_bytes(genome)       # data files the reference's tests hold
    (HERE / "chr2.fa").write_bytes(chr2)

    # ---- known-answer vectors of the reference's own tests, checked against the seam model
    register_index("kat", genome)
    assert count_kmers("kat", [b"AAAA", b"AT", b"TAT", b"CCC", b"NNN", b"TCGT"], 1) == \
        [9, 3, 1, 8, 0, 0]                                   # tests/test_count_kmers.py:21-25
    assert count_kmers_from_sequence("kat", b"AAAAATTTTTATCGAATCGA", [0, 4, 9], [4, 2, 3], 1) == \
        [9, 3, 1]                                            # tests/test_count_kmers.py:37-44
    EXPECTED_CHR1 = [0, 10, 9, 8, 7, 6, 5, 4, 4, 4, 6, 5, 4, 4, 4, 0, 0, 0, 0, 0]
    EXPECTED_CHR2 = [10, 10, 9, 8, 7, 6, 5, 4, 4, 4] + [0] * 20   # tests/test_unique_counts.py:17-21
    for binary in (True, False):
        r = run_search(search, genome, range(4, 11), binary, 15)
        assert r["chr1"]["values"] == EXPECTED_CHR1 and r["chr2"]["values"] == EXPECTED_CHR2

    cases = []

    def add(name, fasta_bytes, lengths, is_binary, batch, note="", **kw):
        res = run_search(search, fasta_bytes, lengths, is_binary, batch, **kw)
        cases.append({
            "name": name, "note": note, "fasta": fasta_bytes.decode("latin-1"),
            "kmer_lengths": list(lengths), "is_binary": is_binary, "batch": batch,
            "use_reverse_complement": kw.get("use_reverse_complement", True),
            "initial_search_length": kw.get("initial_search_length", 0),
            "expected": res,
        })

    # G1: the reference's KAT, both modes
    add("G1_genome_4_10_b15_binary", genome, range(4, 11), True, 15, "tests/test_unique_counts.py")
    add("G1_genome_4_10_b15_linear", genome, range(4, 11), False, 15, "tests/test_unique_counts.py")
    # G2: BASELINE config 1
    add("G2_genome_20_200", genome, range(20, 201), True, 10_000_000, "BASELINE.json configs[0]")
    add("G2_genome_20_200_b7", genome, range(20, 201), True, 7, "same, tiny batches")
    # G3: --norc
    add("G3_genome_4_10_norc", genome, range(4, 11), True, 15, "--norc",
        use_reverse_complement=False)
    add("G3_genome_4_10_norc_linear", genome, range(4, 11), False, 1000, "--norc list mode",
        use_reverse_complement=False)

    # G4: random 9 kbp, two records, N run, soft-masked patch, batch 1000
    rng = np.random.default_rng(20260601)
    a = bytearray(random_dna(rng, 6000))
    a[2500:2600] = b"N" * 100
    a[4000:4200] = bytes(a[4000:4200]).lower()
    b = bytearray(random_dna(rng, 3000))
    b[1000:1400] = a[100:500]                  # shared 400-mer between records
    b[2000:2300] = bytes(a[700:1000]).translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]  # rc copy
    g4 = fasta_text([("r1 some description", bytes(a)), ("r2", bytes(b))])
    add("G4_random_8_60_b1000", g4, range(8, 61), True, 1000,
        "N run sits at 2500-2599; batch boundaries at multiples of 1000 keep every N outside any "
        "lookahead window that is followed by more data, so SURVEY A.3(1) does not bite")
    add("G4_random_8_60_whole", g4, range(8, 61), True, 10_000_000)
    add("G4_random_list_12_20_30", g4, [12, 20, 30], False, 1000, "list mode")
    add("G4_random_list_30_12", g4, [30, 12], False, 10_000_000, "list mode, descending order")
    add("G4_random_fixed_16", g4, [16], False, 2048, "fixed k")

    # G5: tandem repeats
    rng = np.random.default_rng(20260602)
    g5 = fasta_text([("t1", tandem_dna(rng, 8000))])
    add("G5_tandem_8_120_b1500", g5, range(8, 121), True, 1500)
    add("G5_tandem_20_255", g5, range(20, 256), True, 10_000_000, "uint8 upper edge")

    # G6: list-mode tail truncation quirk
    rng = np.random.default_rng(20260603)
    g6 = fasta_text([("q", random_dna(rng, 30))])
    add("G6_tail_linear_8", g6, [8], False, 1000, "search.py:590 slice truncation at record end")
    add("G6_tail_binary_8_9", g6, range(8, 10), True, 1000, "binary mode gives 0 at the tail")

    # G8: uint16 / uint32 selection and --initial-search-length invariance
    rng = np.random.default_rng(20260604)
    g8 = fasta_text([("w", tandem_dna(rng, 3000))])
    add("G8_uint16_20_300", g8, range(20, 301), True, 700, "kmax > 255 -> uint16")
    add("G8_initial_len_24", g8, range(20, 201), True, 700, "-l 24", initial_search_length=24)
    add("G8_initial_len_0", g8, range(20, 201), True, 700, "default midpoint")

    # G9: ambiguity characters -- batch large enough that no lookahead window holds an N
    rng = np.random.default_rng(20260605)
    c = bytearray(random_dna(rng, 2500))
    for s, e in ((0, 7), (300, 301), (640, 700), (1200, 1203), (2490, 2500)):
        c[s:e] = b"N" * (e - s)
    c[900:904] = b"nnnn"
    g9 = fasta_text([("n1", bytes(c)), ("n2", bytes(c[500:1500]))])
    add("G9_ambiguous_6_40_whole", g9, range(6, 41), True, 10_000_000, "single epilogue segment")
    add("G9_ambiguous_fixed_24", g9, [24], False, 10_000_000, "upper-case N only, list mode")
    # The lookahead quirk: with batch 640 the N run 640-699 starts exactly at a
    # batch boundary, inside the unmasked lookahead of segment 0.  Kept as DOCUMENTATION of the
    # reference's batch-dependent output under the seam model; the engine implements A.2.
    add("G9_ambiguous_6_40_b640", g9, range(6, 41), True, 640,
        "N run 640-699 starts at a batch boundary, but every position before it is unique well before it reaches the "
        "run: the unmasked lookahead (search.py:751-753) changes nothing here")

    # G10: the lookahead quirk made visible (newmap/search.py:751-753).  The 40 bases in front of the N
    # run are a copy of bases 100-139, so their k-mers stay repeated until they reach the run.  With batch 640 the run
    # lies in the UNMASKED lookahead of segment 0: the reference's probes then contain N's, and under the seam model
    # (every non-ACGT byte one letter) such a k-mer occurs once -- the reference reports a length there, and what it
    # reports depends on --kmer-batch-size.  With one segment per record the mask covers the run and the same positions
    # are 0 (what the engine implements).  Both outputs are kept; the test counts the difference.
    rng = np.random.default_rng(20260607)
    c = bytearray(random_dna(rng, 2500))
    c[600:640] = c[100:140]
    c[640:700] = b"N" * 60
    g10 = fasta_text([("q1", bytes(c))])
    add("G10_lookahead_6_40_whole", g10, range(6, 41), True, 10_000_000, "one segment: the closed form of Appendix A.2")
    add("G10_lookahead_6_40_b640_quirk", g10, range(6, 41), True, 640,
        "reference output depends on --kmer-batch-size here (lookahead is not masked)")

    (HERE / "golden_search.json").write_text(json.dumps({"cases": cases}, indent=0))

    # ---- G7a: update_upper_search_bound on masks (tests/test_upper_search_bound_truncation.py)
    rng = np.random.default_rng(20260606)
    ub_cases = []
    fixed = [
        ([1, 1, 0, 0, 0, 1, 1], 5, 0), ([1, 0, 0, 0, 0, 0, 1], 4, 0),
        ([0, 0, 0, 0, 0, 1, 1, 0, 1], 4, 0), ([1, 1, 0, 0, 1, 0, 0, 0, 0, 0], 4, 0),
        ([0, 0, 1, 1], 4, 0), ([1, 1, 0, 0], 4, 0),
        ([0, 0, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0], 500, 0), ([0, 0, 0, 0], 50, 0),
        ([0, 0, 0, 0, 0, 0], 4, 1), ([1, 1, 1, 1], 50, 0), ([1, 1, 1, 1], 50, 49),
        ([1, 1, 0, 0, 0, 0], 50, 49), ([1, 1, 0, 0, 0, 0], 50, 3), ([1, 1, 0, 0, 0, 0], 5, 2),
    ]
    for _ in range(40):
        n = int(rng.integers(1, 40))
        kmax = int(rng.integers(1, 20))
        extra = int(rng.integers(0, kmax))
        fixed.append((rng.integers(0, 2, n).tolist() if rng.random() < 0.8 else [0] * n, kmax, extra))
    for mask, kmax, extra in fixed:
        m = np.array(mask, dtype=bool)
        ub = np.full(m.size, kmax)
        search.update_upper_search_bound(ub, m, kmax, m.size + extra)
        ub_cases.append({"mask": mask, "kmax": kmax, "buffer_len": m.size + extra,
                         "expected": ub.tolist()})
    # too much lookahead -> AssertionError (test_no_ambiguous_too_much_lookahead)
    try:
        search.update_upper_search_bound(np.full(4, 50), np.ones(4, bool), 50, 4 + 50)
        raised = False
    except AssertionError:
        raised = True
    assert raised

    # ---- G7b: sequence_segments (tests/test_sequence_buffer_iter.py)
    seg_cases = []
    texts = {
        "chr2": chr2, "genome": genome,
        "exact": b">chr1\nAAAAATTTTTATCGAATCGA\n", "single_nt": b">chr1\nATCGATCGA\n",
        "semicolon": b";c1 x\nACGT\nAC\n>c2\n\nGG\n>empty\n>c3 y z\nTTTTTTTTTTTT\n",
        "headerless": b"ACGTACGTAC\nGGG\n>late\nCCCCC\n",
        "crlf": b">w\r\nACGTAC\r\nGTACGT\r\n",
    }
    params = [(1000, 0), (5, 2), (100, 4), (16, 10), (12, 4), (25, 2), (11, 2), (20, 0), (7, 6),
              (3, 0), (4, 3)]
    for tname, text in texts.items():
        for length, overlap in params:
            segs = list(fasta.sequence_segments(io.BytesIO(text), length, overlap))
            seg_cases.append({
                "text": text.decode("latin-1"), "name": tname, "length": length,
                "overlap": overlap,
                "expected": [[s.id.decode("latin-1"), s.data.decode("latin-1"), bool(s.epilogue)]
                             for s in segs]})
    (HERE / "golden_host.json").write_text(json.dumps(
        {"upper_bound": ub_cases, "segments": seg_cases,
         "kat": {"count_kmers": {"kmers": ["AAAA", "AT", "TAT", "CCC", "NNN", "TCGT"],
                                 "expected": [9, 3, 1, 8, 0, 0]},
                 "count_from_sequence": {"sequence": "AAAAATTTTTATCGAATCGA", "starts": [0, 4, 9],
                                         "lengths": [4, 2, 3], "expected": [9, 3, 1]},
                 "chr1_4_10": EXPECTED_CHR1, "chr2_4_10": EXPECTED_CHR2}}, indent=0))
    print(f"wrote {len(cases)} search cases, {len(ub_cases)} upper-bound cases, "
          f"{len(seg_cases)} segment cases")


if __name__ == "__main__":
    main()

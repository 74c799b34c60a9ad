"""Seeded synthetic genomes for the measurement harness: the inputs of
BASELINE.json's configs, generated on the spot because neither datasets nor the network exist."""
from __future__ import annotations

from pathlib import Path

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

# GRCh38-like record lengths of config C3 (chr1..22, X, Y)
HUMAN_SHAPED = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
                138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
                83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def uniform_dna(n: int, seed: int) -> np.ndarray:
    """n bases i.i.d. uniform over ACGT (uint8 ASCII)."""
    rng = np.random.default_rng(seed)
    out = np.empty(n, dtype=np.uint8)
    step = 1 << 26
    for s in range(0, n, step):
        m = min(step, n - s)
        out[s:s + m] = _ACGT[rng.integers(0, 4, m, dtype=np.uint8)]
    return out


def tandem_dna(n: int, seed: int, frac: float = 0.5) -> np.ndarray:
    """config C5: alternate unique spacers (50..5000) and tandem arrays (unit 2..200, array
    200..50000 bases) until `frac` of the bases sit in arrays."""
    rng = np.random.default_rng(seed)
    out = np.empty(n, dtype=np.uint8)
    pos = in_arrays = 0
    while pos < n:
        m = min(int(rng.integers(50, 5001)), n - pos)
        out[pos:pos + m] = _ACGT[rng.integers(0, 4, m, dtype=np.uint8)]
        pos += m
        if pos >= n:
            break
        if in_arrays < frac * pos:
            unit = _ACGT[rng.integers(0, 4, int(rng.integers(2, 201)), dtype=np.uint8)]
            m = min(int(rng.integers(200, 50001)), n - pos)
            reps = -(-m // unit.size)
            out[pos:pos + m] = np.tile(unit, reps)[:m]
            pos += m
            in_arrays += m
    return out


def write_fasta(path, records: list[tuple[str, np.ndarray]], width: int = 60):
    """60-column FASTA without a Python loop per line."""
    path = Path(path)
    with open(path, "wb") as fh:
        for name, seq in records:
            fh.write(b">" + name.encode() + b"\n")
            n = seq.size
            full = n // width
            if full:
                block = np.empty((full, width + 1), dtype=np.uint8)
                block[:, :width] = seq[:full * width].reshape(full, width)
                block[:, width] = 10
                fh.write(block.tobytes())
            if n % width:
                fh.write(seq[full * width:].tobytes() + b"\n")
    return path


def config_genome(name: str, scale_mbp: float | None = None) -> list[tuple[str, np.ndarray]]:
    """Records of a BASELINE config: 'c2' (100 Mbp uniform, seed 20260515), 'c3' (24 human-shaped
    records, seeds 20260516+i), 'c5' (1 Gbp, 50 % tandem repeats, seed 20260517).  `scale_mbp`
    shrinks the total size proportionally (tests, quick runs)."""
    if name == "c2":
        n = int((scale_mbp or 100) * 1e6)
        return [("chr1", uniform_dna(n, 20260515))]
    if name == "c3":
        total = sum(HUMAN_SHAPED)
        f = 1.0 if scale_mbp is None else scale_mbp * 1e6 / total
        names = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]
        return [(nm, uniform_dna(max(1000, int(L * f)), 20260516 + i))
                for i, (nm, L) in enumerate(zip(names, HUMAN_SHAPED))]
    if name == "c5":
        n = int((scale_mbp or 1000) * 1e6)
        return [("rep1", tandem_dna(n, 20260517))]
    raise ValueError(f"unknown config {name!r}")

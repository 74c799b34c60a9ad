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


def _mutate(copy: np.ndarray, divergence: float, rng) -> np.ndarray:
    """substitutions at `divergence` of the positions (a different base each), a sprinkle of 1-base deletions"""
    out = copy.copy()
    n = out.size
    k = rng.binomial(n, divergence)
    if k:
        pos = rng.integers(0, n, k)
        out[pos] = _ACGT[(np.searchsorted(_ACGT, out[pos]) + rng.integers(1, 4, k)) & 3]
    d = rng.binomial(n, divergence / 10.0)
    if d:
        keep = np.ones(n, dtype=bool)
        keep[rng.integers(0, n, d)] = False
        out = out[keep]
    return out


def _revcomp(a: np.ndarray) -> np.ndarray:
    return _ACGT[3 - np.searchsorted(_ACGT, a[::-1])]


def human_like_dna(n: int, seed: int) -> np.ndarray:
    """A record shaped like a human chromosome rather than like noise (BASELINE configs[3] stand-in: the GRCh38 file is on no
    box).  On a uniform ACGT background:
      * interspersed repeats, both strands: a 300-base family at 5-20 % divergence from its consensus (~10 % of the
        record) and a 6-kb family, copies truncated at the 5' end, at 2-15 % (~15 %); the consensus sequences depend on
        seed // 100 only, so the records of one genome share them;
      * segmental duplications: a handful of 10-100 kb stretches copied elsewhere in the record at 0-2 % divergence;
      * soft-masking: the repeat copies and random stretches in lower case, about half of the record;
      * upper-case N: 10 kb telomeres, a centromere run of up to 3 Mb (scaled with the record), a few 10-100 kb gaps.
    Lower-case bases are searched like upper-case ones (newmap/search.py:23); list mode drops a position whose k-mer
    holds an N (:593-596)."""
    rng = np.random.default_rng(seed)
    fam = np.random.default_rng(seed // 100 + 7)
    alu = _ACGT[fam.integers(0, 4, 300, dtype=np.uint8)]
    l1 = _ACGT[fam.integers(0, 4, 6000, dtype=np.uint8)]
    out = uniform_dna(n, seed)
    lower = np.zeros(n, dtype=bool)

    def plant(consensus, share, div_lo, div_hi, truncate):
        target, done = int(n * share), 0
        while done < target:
            m = consensus.size if not truncate else int(rng.integers(consensus.size // 10, consensus.size + 1))
            copy = _mutate(consensus[consensus.size - m:], float(rng.uniform(div_lo, div_hi)), rng)
            if rng.integers(0, 2):
                copy = _revcomp(copy)
            at = int(rng.integers(0, max(n - copy.size, 1)))
            m = min(copy.size, n - at)
            out[at:at + m] = copy[:m]
            lower[at:at + m] = True
            done += m

    plant(alu, 0.10, 0.05, 0.20, False)
    plant(l1, 0.15, 0.02, 0.15, True)
    for _ in range(max(1, n // 25_000_000)):                # segmental duplications
        m = int(min(rng.integers(10_000, 100_001), n // 4))
        src, dst = int(rng.integers(0, n - m)), int(rng.integers(0, n - m))
        dup = _mutate(out[src:src + m], float(rng.uniform(0.0, 0.02)), rng)
        out[dst:dst + dup.size] = dup
    masked = int(lower.sum())
    while masked < n // 2:                                  # further soft-masked stretches
        m = int(rng.integers(200, 20_001))
        at = int(rng.integers(0, max(n - m, 1)))
        masked += m - int(lower[at:at + m].sum())
        lower[at:at + m] = True
    out = np.where(lower, out | 0x20, out).astype(np.uint8)
    tel = min(10_000, n // 50)
    out[:tel] = ord("N")
    out[n - tel:] = ord("N")
    cen = min(3_000_000, n // 40)
    at = n // 3
    out[at:at + cen] = ord("N")
    for _ in range(max(1, n // 30_000_000)):
        m = int(min(rng.integers(10_000, 100_001), n // 100))
        at = int(rng.integers(0, max(n - m, 1)))
        out[at:at + m] = ord("N")
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
                fh.write(block.reshape(-1).data)                  # (the array's own buffer: no second copy of a 250 MB record)
            if n % width:
                fh.write(seq[full * width:].tobytes() + b"\n")
    return path


def _records_in_parallel(names, make) -> list[tuple[str, np.ndarray]]:
    """the 24 records of a genome, each from its own seed: generated by a few threads (numpy's generators and gathers run
    without the interpreter lock), same bytes as one after the other"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4))
    with ThreadPoolExecutor(workers) as pool:
        seqs = list(pool.map(lambda t: make(t[0], t[1]), enumerate(HUMAN_SHAPED)))
    return list(zip(names, seqs))


def config_genome(name: str, scale_mbp: float | None = None) -> list[tuple[str, np.ndarray]]:
    """Records of a BASELINE config: 'c2' (100 Mbp uniform, seed 20260515), 'c3' (24 human-shaped
    records, seeds 20260516+i), 'c5' (1 Gbp, 50 % tandem repeats, seed 20260517), 'hs' (24 records of
    human_like_dna: the stand-in of configs[3]'s GRCh38).  `scale_mbp`
    shrinks the total size proportionally (tests, quick runs)."""
    if name == "c2":
        n = int((scale_mbp or 100) * 1e6)
        return [("chr1", uniform_dna(n, 20260515))]
    if name == "c3":
        total = sum(HUMAN_SHAPED)
        f = 1.0 if scale_mbp is None else scale_mbp * 1e6 / total
        names = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]
        return _records_in_parallel(names, lambda i, L: uniform_dna(max(1000, int(L * f)), 20260516 + i))
    if name == "c5":
        n = int((scale_mbp or 1000) * 1e6)
        return [("rep1", tandem_dna(n, 20260517))]
    if name == "hs":                                        # the human-shaped stand-in of configs[3]: 24 records of human_like_dna
        total = sum(HUMAN_SHAPED)
        f = 1.0 if scale_mbp is None else scale_mbp * 1e6 / total
        names = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]
        return _records_in_parallel(names, lambda i, L: human_like_dna(max(100_000, int(L * f)), 20260600 + i))
    raise ValueError(f"unknown config {name!r}")

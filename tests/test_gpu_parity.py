"""Parity of the HIP engine with the oracle and with the reference's fixtures, through the C-ABI.
Needs a real MI355X:  python -m pytest tests -m gpu"""
import ctypes
import io
import os
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ref_driver as rd  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    from newmap_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no HIP device visible: the engine has no CPU fallback")
    return engine


def _build_index(tmp_path, text: bytes, name="x", seed=12):
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    fa = tmp_path / f"{name}.fa"
    fa.write_bytes(text)
    idx = tmp_path / f"{name}.awfmi"
    generate_fm_index(str(fa), str(idx), 8, seed)
    return fa, idx


def _random_dna(rng, n):
    return bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)])


# ------------------------------------------------------------------ reference fixtures, end to end
def test_write_unique_counts_reproduces_reference_fixtures(tmp_path, golden_search, eng):
    from newmap_amd.search import SearchConfig, write_unique_counts
    whole = {c["name"]: c for c in golden_search}
    for i, c in enumerate(golden_search):
        d = tmp_path / f"case{i}"
        d.mkdir()
        fa, idx = _build_index(d, c["fasta"].encode("latin-1"))
        out = d / "out"
        out.mkdir()
        write_unique_counts(SearchConfig(
            fasta_filepaths=[fa], fmindex_filepaths=[idx], kmer_lengths=c["kmer_lengths"],
            is_binary_search=c["is_binary"], kmer_batch_size=c["batch"], output_directory=out,
            use_reverse_complement=c["use_reverse_complement"],
            initial_search_length=c["initial_search_length"]))
        files = sorted(p.name for p in out.iterdir())
        assert files == sorted(f"{rid}.unique.{e['dtype']}" for rid, e in c["expected"].items()), c["name"]
        for rid, e in c["expected"].items():
            got = np.fromfile(out / f"{rid}.unique.{e['dtype']}", dtype=e["dtype"])
            if "quirk" in c["name"]:
                # DOCUMENTED DIVERGENCE: the reference does not mask the lookahead of a non-final
                # batch (newmap/search.py:751-753), so 39 positions in front of an N run get a length that depends on
                # --kmer-batch-size; the engine gives the batch-independent answer = the one-segment fixture
                same = whole[c["name"].replace("_b640_quirk", "_whole")]["expected"][rid]["values"]
                assert got.tolist() == same, (c["name"], rid)
                diff = np.flatnonzero(got != np.array(e["values"], dtype=got.dtype))
                assert diff.tolist() == list(range(601, 640)) and not got[diff].any() and np.array(e["values"])[diff].all()
                continue
            assert got.tolist() == e["values"], (c["name"], rid)
    eng.close_all()


def test_reference_kat_through_dropin_modules(tmp_path, golden_host, eng):
    """reference tests/test_count_kmers.py, translated 1:1 onto the drop-in module."""
    from newmap_amd._c_newmap_count_kmers import count_kmers, count_kmers_from_sequence
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    golden = os.path.join(os.path.dirname(__file__), "golden", "genome.fa")
    idx = str(tmp_path / "genome.awfmi")
    generate_fm_index(golden, idx, 8, 12)
    k = golden_host["kat"]["count_kmers"]
    assert count_kmers(idx, [s.encode() for s in k["kmers"]], 1) == k["expected"]
    with pytest.raises(TypeError):
        count_kmers(idx, ["AAAA"], 1)
    with pytest.raises(ValueError):
        count_kmers(idx, [b"AAAA", b"", b"TAT"], 1)
    k = golden_host["kat"]["count_from_sequence"]
    assert count_kmers_from_sequence(idx, k["sequence"].encode(), k["starts"], k["lengths"], 1) == k["expected"]
    with pytest.raises(IndexError):
        count_kmers_from_sequence(idx, b"ACGT", [2], [3], 1)
    with pytest.raises(IndexError):
        count_kmers_from_sequence(idx, b"ACGT", [-1], [3], 1)
    with pytest.raises(ValueError):
        count_kmers_from_sequence(idx, b"ACGT", [0, 1], [3], 1)
    with pytest.raises(TypeError):
        count_kmers_from_sequence(idx, b"ACGT", [0.5], [3], 1)
    with pytest.raises(OverflowError):
        count_kmers(idx, [b"A"], 256)
    with pytest.raises(OSError):
        count_kmers(str(tmp_path / "missing.awfmi"), [b"A"], 1)
    bad = tmp_path / "bad.awfmi"
    bad.write_bytes(b"not an index" * 100)
    with pytest.raises(OSError):
        count_kmers(str(bad), [b"A"], 1)
    eng.close_all()


def test_upper_bound_cases_on_device(tmp_path, golden_host, eng):
    _, idx = _build_index(tmp_path, b">x\nACGTACGTAC\n")
    with eng.Index(idx, 0) as ix:
        for c in golden_host["upper_bound"]:
            mask = np.array(c["mask"], dtype=bool)
            seq = bytes(np.where(mask, ord("N"), ord("A")).astype(np.uint8)) + b"C" * (c["buffer_len"] - mask.size)
            got = ix.upper_bound_segment(seq, mask.size, c["kmax"])
            assert got.tolist() == c["expected"], c
        with pytest.raises(AssertionError):
            ix.upper_bound_segment(b"N" * 4 + b"A" * 50, 4, 50)


# ------------------------------------------------------------------ seeded inputs vs the oracle
@pytest.fixture(scope="module")
def mixed_genome(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("mixed")
    rng = np.random.default_rng(4242)
    r1 = bytearray(_random_dna(rng, 300_000))
    unit = _random_dna(rng, 13)
    r1[50_000:80_000] = (unit * 3000)[:30_000]                 # long tandem array
    r1[120_000:120_400] = b"N" * 400
    r1[200_000:203_000] = bytes(r1[10_000:13_000]).lower()     # soft-masked duplicate
    r1[250_000:250_001] = b"R"                                  # lone IUPAC code
    r2 = _random_dna(rng, 80_000) + bytes(r1[1000:9000]) + \
        bytes(r1[20_000:24_000]).translate(bytes.maketrans(b"ACGTacgt", b"TGCAtgca"))[::-1]
    text = b">one desc\n" + b"\n".join(bytes(r1[i:i + 70]) for i in range(0, len(r1), 70)) + \
           b"\n>two\n" + r2 + b"\n"
    fa, idx = _build_index(tmp, text, "mixed")
    return {"r1": bytes(r1), "r2": r2, "fa": fa, "idx": idx,
            "oracle": rd.OracleIndex([bytes(r1), r2])}


@pytest.mark.parametrize("kmin,kmax,rc", [(20, 200, True), (24, 150, True), (20, 255, True),
                                          (8, 60, False), (20, 400, True), (30, 70000, True)])
def test_min_unique_equals_oracle(mixed_genome, eng, kmin, kmax, rc):
    g = mixed_genome
    # very long ranges: keep the oracle's O(k) text comparisons affordable
    recs = (g["r1"], g["r2"]) if kmax <= 1000 else (g["r1"][49_000:56_000], g["r2"][:5000])
    wants = [rd.closed_form_min_unique(rec, g["oracle"], kmin, kmax, rc) for rec in recs]      # (once: the same for every table set-up)
    for seed in (None, 0, 6):
        with eng.Index(g["idx"], 0, seed) as ix:
            for rec, want in zip(recs, wants):
                got, n_amb = ix.min_unique_segment(rec, len(rec), kmin, kmax, rc)
                assert got.dtype == want.dtype
                assert np.array_equal(got, want), (seed, kmin, kmax, rc)
                assert n_amb == sum(ch not in b"ACGTacgt" for ch in rec)


def test_batch_independence_and_segment_api(mixed_genome, eng):
    """segments with kmax-1 lookahead, stitched, equal the whole-record answer (Appendix A.2)"""
    g = mixed_genome
    rec = g["r1"]
    kmin, kmax = 20, 200
    with eng.Index(g["idx"], 0) as ix:
        whole, _ = ix.min_unique_segment(rec, len(rec), kmin, kmax)
        for batch in (1000, 65_536, 99_999):
            parts = []
            for seg in rd.record_segments(b"one", rec, batch + kmax - 1, kmax - 1):
                n = rd.num_kmers_of(seg, kmax)
                arr, _ = ix.min_unique_segment(seg.data, n, kmin, kmax)
                parts.append(arr)
            assert np.array_equal(np.concatenate(parts), whole), batch


@pytest.mark.parametrize("ks,rc", [([36], True), ([100], True), ([12, 20, 30], True), ([30, 12], True),
                                   ([24], False), ([300, 20], True), ([15], True), ([14], True), ([60], True), ([61], True),
                                   ([124], True), ([125], True), ([1000], True), ([20, 36, 100], True), ([36, 20, 50], True),
                                   ([100, 24], True), ([15, 16], True), ([124, 300], True)])
def test_fixed_k_equals_oracle(mixed_genome, eng, ks, rc):
    g = mixed_genome
    kmax = max(ks)
    dtype, _ = rd.output_dtype(kmax)
    with eng.Index(g["idx"], 0) as ix:
        for rec in (g["r1"][:150_000], g["r2"]):
            # the lone 'R' (not upper-case N) is outside the compared prefix of r1: list mode on
            # other ambiguity codes is a documented divergence
            for batch in (len(rec), 40_000):
                got, want = [], []
                for seg in rd.record_segments(b"r", rec, batch + kmax - 1, kmax - 1):
                    n = rd.num_kmers_of(seg, kmax)
                    w, _ = rd.linear_search_segment(g["oracle"], seg, ks, kmax, dtype, rc)
                    a, amb_a = ix.fixed_k_segment(seg.data, n, ks, rc)
                    got.append(a)
                    want.append(w)
                    if rc:                                 # the range / quad kernels (default where they apply) vs the list kernel
                        ix.set_list_via_range(False)
                        b, amb_b = ix.fixed_k_segment(seg.data, n, ks, rc)
                        ix.set_list_via_range(True)
                        assert np.array_equal(a, b) and amb_a == amb_b, (ks, batch)
                assert np.array_equal(np.concatenate(got), np.concatenate(want)), (ks, rc, batch)


def test_counts_equal_oracle(mixed_genome, eng):
    g = mixed_genome
    rng = np.random.default_rng(5)
    seq = g["r2"]
    starts = rng.integers(0, len(seq) - 300, 20_000)
    lens = rng.integers(1, 300, 20_000)
    with eng.Index(g["idx"], 0) as ix:
        got = ix.count_from_sequence(seq, starts, lens)
        assert np.array_equal(got, g["oracle"].count_from_sequence(seq, starts, lens))
        kmers = [seq[s:s + l] for s, l in zip(starts[:500], lens[:500])] + [b"NNNN", b"ACGTN", b"acgt"]
        assert ix.count_kmers(kmers).tolist()[:500] == got[:500].tolist()
        assert ix.count_kmers(kmers).tolist()[500:502] == [0, 0]


def test_zero_count_guard_raises(tmp_path, eng):
    _, idx = _build_index(tmp_path, b">x\n" + b"ACGTTGCAAGGCTTAACCGGATATCGCGAT" * 4 + b"\n", "small", 4)
    with eng.Index(idx, 0) as ix:
        with pytest.raises(RuntimeError, match="not found in the index"):
            ix.min_unique_segment(b"G" * 64, 64, 4, 8)
        with pytest.raises(RuntimeError, match="not found in the index"):
            ix.fixed_k_segment(b"G" * 64, 64, [6])


def test_zero_count_guard_scope(tmp_path, eng):
    """newmap/search.py:699-722 raises when ANY probe of its bisection schedule is absent from the index.  The engine's
    fast paths ask the index about far fewer k-mers, so exactness comes from the record check (csrc/nm_hash.h): a
    segment / record that is, by length and fingerprint, one of the indexed records cannot hold an absent k-mer; any
    other one goes through the exact guard (nm_guard_*), which replays the reference's probe schedule.  A different
    genome, a single substituted base (range mode AND fixed k = 36 on a small genome, where a group of sites is larger
    than a window and the base can fall between the looked-up windows), a chimeric join of two indexed pieces: all
    raise; a piece of an indexed record -- not an indexed record, but free of absent k-mers -- does not, and equals the
    oracle."""
    rng = np.random.default_rng(99)
    genome = _random_dna(rng, 300_000)
    fa, idx = _build_index(tmp_path, b">g\n" + genome + b"\n", "guard")
    with eng.Index(idx, 0) as ix:
        ok, _ = ix.min_unique_segment(genome, len(genome), 20, 200)
        assert ok[:-19].min() >= 20 and ix.guard_segments() == 0          # an indexed record: no guard
        with pytest.raises(RuntimeError, match="not found in the index"):
            ix.min_unique_segment(_random_dna(rng, 100_000), 100_000, 20, 200)
        for at in (1234, 150_001, 299_000):
            snp = bytearray(genome)
            snp[at] = ord("A") if genome[at] != ord("A") else ord("C")
            with pytest.raises(RuntimeError, match="not found in the index"):
                ix.min_unique_segment(bytes(snp), len(snp), 20, 200)
            for ks in ([36], [100], [24, 36]):
                with pytest.raises(RuntimeError, match="not found in the index"):
                    ix.fixed_k_segment(bytes(snp), len(snp), ks)
        chimera = genome[1000:90_000] + genome[200_000:260_000]
        with pytest.raises(RuntimeError, match="not found in the index"):
            ix.min_unique_segment(chimera, len(chimera), 20, 200)
        before = ix.guard_segments()
        piece = genome[50_000:250_000]
        oracle = rd.OracleIndex([genome])
        got, _ = ix.min_unique_segment(piece, len(piece), 20, 200)
        assert ix.guard_segments() == before + 1
        assert np.array_equal(got, rd.closed_form_min_unique(piece, oracle, 20, 200))
        got, _ = ix.fixed_k_segment(piece, len(piece), [36])
        want, _ = rd.linear_search_segment(oracle, rd.Segment(b"p", piece, True, 0), [36], 36, np.uint8)
        assert np.array_equal(got, want)
        # --initial-search-length shapes the schedule the guard replays, never a result
        got2, _ = ix.min_unique_segment(piece, len(piece), 20, 200, initial_search_length=30)
        assert np.array_equal(got2, got if False else rd.closed_form_min_unique(piece, oracle, 20, 200))


def test_record_fingerprints_and_native_driver_guard(tmp_path, eng, monkeypatch):
    """csrc/nm_hash.h end to end: the index lists its records' fingerprints; every path of the device (sites, one lane per
    position, --norc, list mode) leaves the same fingerprint for the same positions, and segments cut at multiples of 64
    join to the record's; the native driver -- parallel and streaming front-ends, any batch -- searches a FASTA that IS
    the indexed genome without a single guard segment, raises for a FASTA with one substituted base (k = 36 list mode on
    a small genome included), and sends records that merely are not indexed ones (a piece; a renamed, reordered file is
    fine) through the guard without raising."""
    from newmap_amd import _lib, engine
    import torch
    rng = np.random.default_rng(123)
    r1 = bytearray(_random_dna(rng, 400_001))
    r1[1000:1040] = b"N" * 40
    r1[70_000:70_100] = bytes(r1[70_000:70_100]).lower()
    r1, r2 = bytes(r1), _random_dna(rng, 12_345)
    text = b">one\n" + b"\n".join(r1[i:i + 70] for i in range(0, len(r1), 70)) + b"\n>two\n" + r2 + b"\n"
    fa, idx = _build_index(tmp_path, text, "fp")
    SW = _lib.NM_STATUS_WORDS
    with eng.Index(idx, 0) as ix:
        lens, fps = ix.records()
        want = sorted([(len(r), engine.fingerprint(r)) for r in (r1, r2)])
        assert sorted(zip(lens.tolist(), fps.tolist())) == want
        fp1 = engine.fingerprint(r1)
        dev = torch.device("cuda", 0)
        seq = torch.frombuffer(bytearray(r1), dtype=torch.uint8).to(dev)
        out = torch.zeros(len(r1), dtype=torch.uint8, device=dev)
        for mode in ("sites", "lanes", "norc", "list", "list3"):
            for batch in (len(r1), 64 * 1000, 64 * 1563):
                st = torch.zeros((len(r1) // batch + 2, SW), dtype=torch.int64, device=dev)
                ix.set_kernel(1 if mode == "lanes" else 0)
                joined = 0
                for j, a in enumerate(range(0, len(r1), batch)):
                    cnt = min(batch, len(r1) - a)
                    kmax = {"list": 36, "list3": 50}.get(mode, 200)
                    seg_len = min(len(r1), a + cnt + kmax - 1) - a
                    if mode.startswith("list"):
                        ix.fixed_k_segment_dev(seq.data_ptr() + a, seg_len, cnt, [36] if mode == "list" else [24, 36, 50], True, 1, out.data_ptr() + a, st.data_ptr() + 8 * SW * j)
                    else:
                        ix.min_unique_segment_dev(seq.data_ptr() + a, seg_len, cnt, 20, 200, mode != "norc", 1, out.data_ptr() + a, st.data_ptr() + 8 * SW * j)
                    torch.cuda.synchronize()
                    joined = (joined + engine.fingerprint_join(0, a, int(st[j, _lib.NM_STATUS_HASH].item()) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
                assert joined == fp1, (mode, batch)
        ix.set_kernel(0)

        def run(fasta, out_name, ks, is_range, batch, **env):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            out_dir = tmp_path / out_name
            out_dir.mkdir(exist_ok=True)
            try:
                return ix.search_fasta(fasta, out_dir, ks, is_range, True, batch), out_dir
            finally:
                for k_ in env:
                    monkeypatch.delenv(k_)

        for env in ({"NEWMAP_AMD_DRIVER_FUSE": "0"}, {"NEWMAP_AMD_STREAMING_DRIVER": "1"}, {}):
            # (the drivers round their working batch down to a multiple of 64 bases: segments start at words of their record and
            #  their fingerprints join, whatever --kmer-batch-size is -- 1000 works like 960; below 64 nothing can be rounded:
            #  the segments do not join and the record goes through the exact guard, which does not raise for an indexed record)
            for batch in (10_000_000, 64 * 777, 1000, 50):
                if not env and batch == 50:
                    continue                                             # (fused into one unit per record: nothing to join)
                before = ix.guard_segments()
                total, out_dir = run(fa, "same", [20, 200], True, batch, **env)
                assert total["positions"] == len(r1) + len(r2)
                assert (ix.guard_segments() == before) == (batch != 50), (env, batch)
        same_one = np.fromfile(tmp_path / "same" / "one.unique.uint8", dtype=np.uint8)
        # a FASTA with the records renamed and reordered is still the indexed genome
        fa2 = tmp_path / "renamed.fa"
        fa2.write_bytes(b">b\n" + r2 + b"\n>a\n" + r1 + b"\n")
        before = ix.guard_segments()
        run(fa2, "renamed", [20, 200], True, 10_000_000)
        assert ix.guard_segments() == before
        assert np.array_equal(np.fromfile(tmp_path / "renamed" / "a.unique.uint8", dtype=np.uint8), same_one)
        # one substituted base: the reference raises, so does the driver -- range mode and k = 36
        snp = bytearray(r1)
        snp[222_222] = ord("A") if r1[222_222] != ord("A") else ord("C")
        fa3 = tmp_path / "snp.fa"
        fa3.write_bytes(b">one\n" + bytes(snp) + b"\n>two\n" + r2 + b"\n")
        for env in ({}, {"NEWMAP_AMD_STREAMING_DRIVER": "1"}):
            for ks, is_range in (([20, 200], True), ([36], False), ([100], False)):
                with pytest.raises(RuntimeError, match="not found in the index"):
                    run(fa3, "snp", ks, is_range, 10_000_000, **env)
        # a piece of a record: guarded, nothing absent, output = the corresponding stretch (away from the piece's end)
        fa4 = tmp_path / "piece.fa"
        fa4.write_bytes(b">piece\n" + r1[100_000:300_000] + b"\n")
        before = ix.guard_segments()
        run(fa4, "piece", [20, 200], True, 10_000_000)
        assert ix.guard_segments() > before
        got = np.fromfile(tmp_path / "piece" / "piece.unique.uint8", dtype=np.uint8)
        assert np.array_equal(got[:150_000], same_one[100_000:250_000])


def test_open_errors(tmp_path, eng):
    with pytest.raises(FileNotFoundError):
        eng.Index(tmp_path / "nope.awfmi", 0)
    junk = tmp_path / "junk.awfmi"
    junk.write_bytes(os.urandom(4096))
    with pytest.raises(OSError):
        eng.Index(junk, 0)
    _, idx = _build_index(tmp_path, b">x\nACGTACGTAC\n")
    with pytest.raises(RuntimeError):
        eng.Index(idx, -1)                      # no CPU path
    with pytest.raises(RuntimeError):
        eng.Index(idx, 99)
    # the switches that cut work out of the kernels (wrong results, timing experiments) are not in this library
    from newmap_amd import _lib
    with eng.Index(idx, 0) as ix:
        for bits in (0x100, 0x200, 0x300):
            with pytest.raises(ValueError, match="measurement build"):
                _lib.raise_for(ix._L.nm_set_option(ix.handle, _lib.NM_OPT_SEED_POLICY, bits))
        for bits in (0, 1, 2, 0x800, 0x1000, 0x2000):     # cache policies and A/B switches: same results
            _lib.raise_for(ix._L.nm_set_option(ix.handle, _lib.NM_OPT_SEED_POLICY, bits))
        with pytest.raises(ValueError):
            ix.set_sweep(3)


def test_indexes_of_one_search_share_the_hbm(tmp_path, eng):
    """newmap/search.py:656-697 sums the counts of every index file: the handles of one search are resident together.  Two
    100 Mbp indexes opened for one search ("auto" tables, equal budgets of the free HBM: engine.cached_indexes) BOTH get
    their quad tables, seed tables and LF blocks; a budget that leaves room for the small tables only gives those; a
    budgeted handle's results are the unbudgeted handle's."""
    from newmap_amd import synth
    recs = [synth.uniform_dna(100_000_000, 41), synth.uniform_dna(100_000_000, 42)]
    paths = []
    for i, r in enumerate(recs):
        fa = tmp_path / f"g{i}.fa"
        synth.write_fasta(fa, [(f"g{i}", r)])
        idx = tmp_path / f"g{i}.awfmi"
        from newmap_amd._c_newmap_generate_index import generate_fm_index
        generate_fm_index(str(fa), str(idx), 8, 12, device=0)
        paths.append(idx)
    a, b = eng.cached_indexes(paths, 0)
    ia, ib = a.info(), b.info()
    for info in (ia, ib):
        assert info["quad_core_length"] >= 13 and info["quad_small_core_length"] >= 8 and info["seed_length"] >= 15 and info["lf_blocks"] == 1
    assert ia["device_bytes"] + ib["device_bytes"] < 200e9
    seg = recs[1][:5_000_199].tobytes()
    want, _ = b.min_unique_segment(seg, 5_000_000, 20, 200)
    eng.close_all()
    with eng.Index(paths[1], 0, "auto", hbm_budget=6 << 30) as tight:       # rank + strand + LF blocks of a 100 Mbp index are ~0.5 GB
        info = tight.info()
        assert info["device_bytes"] < (6 << 30) and 0 < info["quad_core_length"] < ia["quad_core_length"]
        got, _ = tight.min_unique_segment(seg, 5_000_000, 20, 200)
    assert np.array_equal(got, want)


def test_empty_and_tiny_inputs(tmp_path, eng):
    _, idx = _build_index(tmp_path, b">x\nACGTACGTACGGTTAACC\n>y\nNNNN\n>z\nA\n", "tiny", 3)
    with eng.Index(idx, 0) as ix:
        arr, amb = ix.min_unique_segment(b"", 0, 4, 8)
        assert arr.size == 0 and amb == 0
        arr, amb = ix.min_unique_segment(b"NNNN", 4, 4, 8)
        assert arr.tolist() == [0, 0, 0, 0] and amb == 4
        arr, amb = ix.min_unique_segment(b"A", 1, 4, 8)
        assert arr.tolist() == [0] and amb == 0
        assert ix.count_kmers([b"A"]).tolist() == [6]


# ------------------------------------------------------------------ device-resident API + torch
def test_device_pointer_api_matches_host_api(mixed_genome, eng):
    from newmap_amd import _lib
    L = _lib.lib()
    g = mixed_genome
    rec = g["r1"]
    with eng.Index(g["idx"], 0) as ix:
        want, amb = ix.min_unique_segment(rec, len(rec), 20, 200)
        d_seq, d_out, d_st = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        assert L.nm_dev_alloc(0, len(rec), ctypes.byref(d_seq)) == 0
        assert L.nm_dev_alloc(0, len(rec), ctypes.byref(d_out)) == 0
        assert L.nm_dev_alloc(0, 64, ctypes.byref(d_st)) == 0
        buf = np.frombuffer(rec, np.uint8)
        assert L.nm_dev_upload(0, d_seq, buf.ctypes.data, buf.size) == 0
        ix.set_count_steps(True)
        ix.min_unique_segment_dev(d_seq.value, len(rec), len(rec), 20, 200, True, 1, d_out.value, d_st.value)
        assert L.nm_dev_sync(0) == 0
        got = np.zeros(len(rec), np.uint8)
        st = np.zeros(8, np.uint64)
        assert L.nm_dev_download(0, got.ctypes.data, d_out, got.size) == 0
        assert L.nm_dev_download(0, st.ctypes.data, d_st, 64) == 0
        assert np.array_equal(got, want)
        assert int(st[0]) == amb and int(st[1]) == 0
        assert int(st[7]) == len(rec) - amb and int(st[3]) > 0 and int(st[4]) >= int(st[3])
        ix.set_count_steps(False)
        # a segment that starts at an odd device address takes the byte-wise encode kernel
        off = 12_345
        sub = rec[off:]
        want_sub, _ = ix.min_unique_segment(sub, len(sub), 20, 200)
        ix.min_unique_segment_dev(d_seq.value + off, len(sub), len(sub), 20, 200, True, 1, d_out.value, d_st.value)
        assert L.nm_dev_sync(0) == 0
        got_sub = np.zeros(len(sub), np.uint8)
        assert L.nm_dev_download(0, got_sub.ctypes.data, d_out, got_sub.size) == 0
        assert np.array_equal(got_sub, want_sub)
        # an output buffer at an odd address: the quad kernel falls back from 4-byte to element-wise stores
        sub = rec[:len(rec) - 3]
        want_sub, _ = ix.min_unique_segment(sub, len(sub), 20, 200)
        ix.min_unique_segment_dev(d_seq.value, len(sub), len(sub), 20, 200, True, 1, d_out.value + 3, d_st.value)
        assert L.nm_dev_sync(0) == 0
        got_sub = np.zeros(len(sub), np.uint8)
        assert L.nm_dev_download(0, got_sub.ctypes.data, ctypes.c_void_p(d_out.value + 3), got_sub.size) == 0
        assert np.array_equal(got_sub, want_sub)
        for p in (d_seq, d_out, d_st):
            L.nm_dev_free(0, p)


def test_torch_tensors_share_the_runtime(mixed_genome):
    """bench.py and the multi-GPU driver hand torch tensors' data_ptr to the C-ABI: both must sit
    on one HIP runtime (torch is imported first, see newmap_amd/_lib.py)."""
    import torch
    from newmap_amd import engine
    g = mixed_genome
    rec = g["r2"]
    with engine.Index(g["idx"], 0) as ix:
        want, _ = ix.min_unique_segment(rec, len(rec), 24, 150)
        seq = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to("cuda:0")
        out = torch.empty(len(rec), dtype=torch.uint8, device="cuda:0")
        st = torch.zeros(8, dtype=torch.int64, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        ix.min_unique_segment_dev(seq.data_ptr(), len(rec), len(rec), 24, 150, True, 1, out.data_ptr(),
                                  st.data_ptr(), stream)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)


def test_segments_on_several_streams_overlap_safely(tmp_path):
    """The handle keeps one set of launch scratch (encoded words, need bitmap, probe results, counters, side stream) per
    caller stream -- a LANE (nm_engine.hip nm_lane_for) -- so segments launched on different streams may overlap.
    A tandem-rich genome (side-stream probes, resolve walks) and a uniform one, cut into segments of uneven sizes and
    dealt over 1, 2, 3 and 6 streams (6 > lanes: the least recently used lane changes hands): every position equals the
    one-stream result, which equals the oracle; list mode likewise; per-segment status rows keep their own counts."""
    from newmap_amd._lib import NM_STATUS_WORDS as SW      # one status row per segment
    import torch
    from newmap_amd import engine, synth
    tandem = synth.tandem_dna(1_500_000, 77).tobytes()
    uniform = synth.uniform_dna(700_000, 78).tobytes()
    text = b">t\n" + tandem + b"\n>u\n" + uniform + b"\n"
    fa, idx = _build_index(tmp_path, text, "lanes")
    oracle = rd.OracleIndex([tandem, uniform], max_depth=300)     # (tandem arrays: suffixes ordered by their first 300 symbols, counts up to kmax exact -- tests/test_oracle_golden.py pins it to the full order)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for rec, (kmin, kmax) in ((tandem, (20, 255)), (uniform, (20, 200)), (tandem, (24, 60))):
        want = rd.closed_form_min_unique(rec, oracle, kmin, kmax)
        n = len(rec)
        cuts = np.concatenate(([0], np.sort(rng.choice(np.arange(1, n), 22, replace=False)), [n]))
        cuts[1:6] = cuts[0] + np.arange(1, 6) * 70_001 if n > 500_000 else cuts[1:6]        # a few launches big enough for the side-stream probes
        cuts = np.unique(cuts)
        seq = torch.frombuffer(bytearray(rec), dtype=torch.uint8).to(dev)
        with engine.Index(idx, 0) as ix:
            results = []
            for n_streams in (1, 2, 6):
                streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
                out = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
                st = torch.zeros((len(cuts) - 1, SW), dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                for rounds in range(2):                                   # second round: every lane is reused while warm
                    for j, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                        a, b = int(a), int(b)
                        seg_len = min(n, b + kmax - 1) - a
                        cnt = b - a if b < n else seg_len
                        ix.min_unique_segment_dev(seq.data_ptr() + a, seg_len, cnt, kmin, kmax, True, 1, out.data_ptr() + a,
                                                  st.data_ptr() + 8 * SW * j, streams[(j + rounds) % n_streams].cuda_stream)
                torch.cuda.synchronize()
                results.append((out.cpu().numpy(), st.cpu().numpy()))
            assert np.array_equal(results[0][0], want), (kmin, kmax)
            for got, status in results[1:]:
                assert np.array_equal(got, want), (kmin, kmax)
                assert np.array_equal(status[:, [0, 1, 7]], results[0][1][:, [0, 1, 7]])      # ambiguous, errors, positions searched
            # list mode over the lanes
            ks = [kmin + 4]
            want_l = None
            for n_streams in (1, 3):
                streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
                out = torch.zeros(n, dtype=torch.uint8, device=dev)
                st = torch.zeros((len(cuts) - 1, SW), dtype=torch.int64, device=dev)
                torch.cuda.synchronize()
                for j, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                    a, b = int(a), int(b)
                    seg_len = min(n, b + ks[0] - 1) - a
                    cnt = b - a if b < n else seg_len
                    ix.fixed_k_segment_dev(seq.data_ptr() + a, seg_len, cnt, ks, True, 1, out.data_ptr() + a, st.data_ptr() + 8 * SW * j,
                                           streams[j % n_streams].cuda_stream)
                torch.cuda.synchronize()
                if want_l is None:
                    want_l = out.cpu().numpy()
                    ref, _ = ix.fixed_k_segment(rec, n, ks)
                    assert np.array_equal(want_l, ref)
                else:
                    assert np.array_equal(out.cpu().numpy(), want_l)


# ------------------------------------------------------------------ larger sizes: properties
def test_large_random_genome_properties(tmp_path, eng):
    """BASELINE configs[1] at FULL size (100 Mbp uniform genome, the bench workload): table
    independence, batch independence, and minimality re-checked through the count seam."""
    from newmap_amd import synth
    rng = np.random.default_rng(20260515)
    rec = synth.config_genome("c2")[0][1].tobytes()
    fa, idx = _build_index(tmp_path, b">chr1\n" + rec + b"\n", "big")
    kmin, kmax = 20, 200
    with eng.Index(idx, 0) as ix:                         # automatic tables: the quad kernel, as in bench.py
        whole, amb = ix.min_unique_segment(rec, len(rec), kmin, kmax)
        assert amb == 0 and ix.info()["last_range_kernel"] == 5
        ix.set_force_big(True)                                # the > 2^31-row instantiations, at full size
        big, _ = ix.min_unique_segment(rec, len(rec), kmin, kmax)
        ix.set_force_big(False)
        assert np.array_equal(big, whole)
        parts = [ix.min_unique_segment(s.data, rd.num_kmers_of(s, kmax), kmin, kmax)[0]
                 for s in rd.record_segments(b"c", rec, 10_000_000 + kmax - 1, kmax - 1)]
        assert np.array_equal(np.concatenate(parts), whole)
        # tail: the last kmin-1 positions cannot hold a k-mer of length kmin
        assert not whole[-(kmin - 1):].any() and whole[:-(kmin - 1)].min() >= kmin
        # minimality through the compat seam: total count at the reported length is 1,
        # and (when above kmin) the one-shorter k-mer is not unique
        comp = bytes.maketrans(b"ACGT", b"TGCA")
        rcrec = rec.translate(comp)[::-1]
        pos = rng.integers(0, len(rec) - kmax, 50_000)
        k = whole[pos].astype(np.int64)
        assert (k > 0).all()
        tot = ix.count_from_sequence(rec, pos, k) + ix.count_from_sequence(rcrec, len(rec) - pos - k, k)
        assert (tot == 1).all()
        longer = k > kmin
        tot2 = ix.count_from_sequence(rec, pos[longer], k[longer] - 1) + \
            ix.count_from_sequence(rcrec, len(rec) - pos[longer] - (k[longer] - 1), k[longer] - 1)
        assert (tot2 > 1).all()
    with eng.Index(idx, 0, "auto-small") as small:          # the one-shot CLI's tables: same kernels, <= 20 GB
        info = small.info()
        assert info["seed_length"] == 15 and info["quad_core_length"] == 13 and info["device_bytes"] < 21e9
        got, _ = small.min_unique_segment(rec[:20_000_199], 20_000_000, kmin, kmax)
        assert small.info()["last_range_kernel"] == 5 and np.array_equal(got, whole[:20_000_000])
    with eng.Index(idx, 0, 12) as ix12:                   # the reference's default seed length, simple kernel
        seed12, _ = ix12.min_unique_segment(rec, len(rec), kmin, kmax)
        assert ix12.info()["last_range_kernel"] == 1 and np.array_equal(seed12, whole)
    with eng.Index(idx, 0, 0) as ix0:
        no_seed, _ = ix0.min_unique_segment(rec[:3_000_000 + kmax], 3_000_000, kmin, kmax)
        assert np.array_equal(no_seed, whole[:3_000_000])
    # and the oracle itself on a prefix (SA counter, reference schedule in C)
    oracle = rd.OracleIndex([rec])
    sample = 200_000
    want, _, _ = rd.ref_binary_search_segment_c(oracle, rec[:sample + kmax - 1], sample, kmin, kmax, fm=False)
    assert np.array_equal(want.astype(np.uint8), whole[:sample])


def _seam_minimality(ix, rec: bytes, out: np.ndarray, kmin: int, kmax: int, rng, samples: int):
    """re-derive sampled elements through the count seam (an independent kernel, k_count, and the strand blocks): at the
    reported length the both-strand count is 1, one base shorter (if allowed) it is not; a 0 means the kmax-mer is
    repeated (or the position cannot hold kmin bases)"""
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rcrec = rec.translate(comp)[::-1]
    n = len(rec)
    pos = rng.integers(0, n - kmax, samples)
    k = out[pos].astype(np.int64)
    found = k > 0
    p1, k1 = pos[found], k[found]
    tot = ix.count_from_sequence(rec, p1, k1) + ix.count_from_sequence(rcrec, n - p1 - k1, k1)
    assert (tot == 1).all()
    longer = k1 > kmin
    tot2 = ix.count_from_sequence(rec, p1[longer], k1[longer] - 1) + \
        ix.count_from_sequence(rcrec, n - p1[longer] - (k1[longer] - 1), k1[longer] - 1)
    assert (tot2 > 1).all()
    p0 = pos[~found]
    if p0.size:
        kk = np.full(p0.size, kmax, dtype=np.int64)
        tot0 = ix.count_from_sequence(rec, p0, kk) + ix.count_from_sequence(rcrec, n - p0 - kmax, kk)
        assert (tot0 > 1).all()
    return int(found.sum()), int(p0.size)


def _oracle_windows(records, windows, kmin: int, kmax: int):
    """ORACLE contact at full size.  `windows` = [(segment bytes incl. lookahead, engine output of its first positions)].
    The claims an output makes about totals (oracle/ref_driver.py closed_form_claims: total == 1 at the reported length,
    >= 2 one base shorter, >= 2 at U_p where 0 is reported -- together they pin the closed form of SURVEY.md Appendix
    A.2 exactly) are checked against totals counted by the oracle's scan counter (or_scan_counts: ONE pass over all the
    records of the genome with a hash of the queries' first kmin bases, forward + reverse complement; no index, no
    suffix array -- nothing of the engine, nothing that a 6 G-symbol text rules out)."""
    blob, starts, lens, rels = [], [], [], []
    base = 0
    for seg, out in windows:
        st, ln, rel = rd.closed_form_claims(seg, out, kmin, kmax)
        blob.append(seg)
        starts.append(st + base)
        lens.append(ln)
        rels.append(rel)
        base += len(seg)
    starts, lens, rels = np.concatenate(starts), np.concatenate(lens), np.concatenate(rels)
    tot = rd.scan_total_counts(records, b"".join(blob), starts, lens, kmin)
    assert ((tot == 1) == (rels == 0)).all(), "a reported length is not the least unique one"
    assert (tot[rels == 1] >= 2).all()
    return int(starts.size)


def test_config3_full_size_properties(tmp_path, eng, monkeypatch):
    """BASELINE configs[2] at FULL size -- 3.09 Gbp, 24 human-shaped records, 24:150, index built on the device (6.18 G BWT
    rows: every kernel runs in its > 2^31-row instantiation) -- and the north-star range 20:200 on the same index.
    No CPU oracle holds a suffix array of the 6 G-symbol both-strand text within this suite's time; the ORACLE is brought in
    by its scan counter instead (_oracle_windows: three windows of 300 k positions per range, every claim of the output
    checked against totals counted over all 24 records).  Beside it, the properties the
    domain offers: (1) minimality of sampled elements re-derived through the count seam, (2) batch independence
    (10 M launches == a 100 M launch, on its first 40 M), (3) the sites == the one-lane-per-position kernel on a 10 M stretch (independent
    schedules of the arithmetic), with either quad table and with the coarse probes forced, (4) structural facts (the
    last kmin-1 positions are 0; every element is 0 or in [kmin, kmax])."""
    from newmap_amd import synth
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    recs = synth.config_genome("c3")
    fa = tmp_path / "c3.fa"
    synth.write_fasta(fa, recs)
    idx = tmp_path / "c3.awfmi"
    generate_fm_index(str(fa), str(idx), 8, 12, device=0)
    rng = np.random.default_rng(31)
    chr1, chr21 = recs[0][1].tobytes(), recs[20][1].tobytes()
    all_records = [r[1] for r in recs]                     # (the oracle's scan counter reads them all)
    monkeypatch.setenv("NEWMAP_AMD_COARSE_MIN", "0")
    with eng.Index(idx, 0) as ix:
        ix.set_segment_guard(False)        # (pieces of records by the hundred million: the record check has its own tests)
        info = ix.info()
        assert info["bwt_length"] > 2 ** 32 and info["quad_core_length"] == 15
        for kmin, kmax in ((24, 150), (20, 200)):
            whole, amb = ix.min_unique_segment(chr1[:100_000_000 + kmax - 1], 100_000_000, kmin, kmax)
            assert amb == 0 and ix.info()["last_range_kernel"] == 5
            assert ((whole == 0) | ((whole >= kmin) & (whole <= kmax))).all()
            parts = [ix.min_unique_segment(chr1[o:o + 10_000_000 + kmax - 1], 10_000_000, kmin, kmax)[0] for o in range(0, 40_000_000, 10_000_000)]
            assert np.array_equal(np.concatenate(parts), whole[:40_000_000])
            found, none = _seam_minimality(ix, chr1[:100_000_000 + kmax], whole, kmin, kmax, rng, 30_000)
            assert found > 29_000
            small, _ = ix.min_unique_segment(chr21, len(chr21), kmin, kmax)          # a whole record: the tail rule
            assert not small[-(kmin - 1):].any() and small[:-(kmin - 1)].min() >= kmin
            _seam_minimality(ix, chr21, small, kmin, kmax, rng, 20_000)
            # the oracle on windows of this index's own genome: start and inside of chr1, the end of chr21 (tail rule included)
            W = 300_000
            n_claims = _oracle_windows(all_records, [(chr1[:W + kmax - 1], whole[:W]),
                                                     (chr1[77_000_000:77_000_000 + W + kmax - 1], whole[77_000_000:77_000_000 + W]),
                                                     (chr21[-W:], small[-W:])], kmin, kmax)
            assert n_claims >= 3 * W - 3 * kmax
            # independent schedules on a 20 M stretch
            sub = chr1[40_000_000:50_000_000 + kmax - 1]
            ix.set_kernel(1)
            ix.set_repeat_probes(False)
            plain, _ = ix.min_unique_segment(sub, 10_000_000, kmin, kmax)
            ix.set_kernel(0)
            ix.set_repeat_probes(True)
            assert np.array_equal(plain, whole[40_000_000:50_000_000])
            for table in (1, 2):
                ix.set_site_table(table)
                got, _ = ix.min_unique_segment(sub, 10_000_000, kmin, kmax)
                assert np.array_equal(got, plain), (kmin, table)
            ix.set_site_table(0)
        # BASELINE configs[3]'s mode (fixed-k list mode, k = 36 and k = 100; the GRCh38 file itself is on no box) at this
        # size: the sites route == the list kernel, and a list of several lengths too
        sub = chr1[40_000_000:50_000_000]
        for ks in ([36], [100], [24, 36, 50, 100]):
            ix.set_list_via_range(True)
            a, _ = ix.fixed_k_segment(sub, len(sub), ks)
            assert ix.info()["last_range_kernel"] == 5
            ix.set_list_via_range(False)
            b, _ = ix.fixed_k_segment(sub, len(sub), ks)
            ix.set_list_via_range(True)
            assert np.array_equal(a, b), ks
            assert (a[:-max(ks)] >= min(ks)).all()
    monkeypatch.setenv("NEWMAP_AMD_COARSE", "2")            # coarse probes forced (they find nothing to settle here)
    with eng.Index(idx, 0, "auto-small") as ix:             # the one-shot CLI's tables on the same index
        got, _ = ix.min_unique_segment(chr1[:30_000_000 + 199], 30_000_000, 20, 200)
    assert np.array_equal(got, whole[:30_000_000])


def test_config5_full_size_properties(tmp_path, eng, monkeypatch):
    """BASELINE configs[4] at FULL size -- 1 Gbp, 50 % tandem repeats, 20:255, index built on the device: the probes settle
    about half of the positions (of a 60 M launch), and the elements satisfy the count-seam properties (unique at the reported length, not
    one base shorter; a 0 = the 255-mer is repeated); 10 M launches == the 60 M launch on its first 30 M; the sites == the one-lane-per-
    position kernel without probes on a stretch; coarse probes forced == not forced."""
    from newmap_amd import synth
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    rec = synth.config_genome("c5")[0][1]
    fa = tmp_path / "c5.fa"
    synth.write_fasta(fa, [("rep1", rec)])
    idx = tmp_path / "c5.awfmi"
    generate_fm_index(str(fa), str(idx), 8, 12, device=0)
    rec = rec.tobytes()
    rng = np.random.default_rng(32)
    kmin, kmax = 20, 255
    n = 60_000_000
    with eng.Index(idx, 0) as ix:
        ix.set_segment_guard(False)                          # (pieces of the record: the record check has its own tests)
        ix.set_count_steps(True)                             # (the probes' tally of settled positions is kept by the counter build)
        ix.set_sweep(True)                                   # (probes, open-word lists from the first launch on: the default waits until the handle has met open positions)
        whole, amb = ix.min_unique_segment(rec[:n + kmax - 1], n, kmin, kmax)
        assert amb == 0 and ix.info()["last_range_kernel"] == 5
        zeros = np.count_nonzero(whole == 0) / n
        assert 0.3 < zeros < 0.7
        assert ix.probe_tally()["settled"] > 0.8 * zeros * n
        ix.set_count_steps(False)
        parts = [ix.min_unique_segment(rec[o:o + 10_000_000 + kmax - 1], 10_000_000, kmin, kmax)[0] for o in range(0, 30_000_000, 10_000_000)]
        assert np.array_equal(np.concatenate(parts), whole[:30_000_000])
        found, none = _seam_minimality(ix, rec[:n + kmax], whole, kmin, kmax, rng, 40_000)
        assert found > 10_000 and none > 10_000
        # the oracle on three windows (tandem arrays, their ends, spacers): totals from one scan of the whole 1 Gbp record
        W = 150_000
        wins = [(rec[o:o + W + kmax - 1], whole[o:o + W]) for o in (0, 30_000_000, 57_700_000)]
        _oracle_windows([rec], wins, kmin, kmax)
        sub = rec[30_000_000:34_000_000 + kmax - 1]
        ix.set_kernel(1)
        ix.set_repeat_probes(False)
        plain, _ = ix.min_unique_segment(sub, 4_000_000, kmin, kmax)
        ix.set_kernel(0)
        ix.set_repeat_probes(True)
        assert np.array_equal(plain, whole[30_000_000:34_000_000])
        tail, _ = ix.min_unique_segment(rec[-20_000_000:], 20_000_000, kmin, kmax)   # the end of the record
        assert not tail[-(kmin - 1):].any()
    monkeypatch.setenv("NEWMAP_AMD_COARSE", "2")
    monkeypatch.setenv("NEWMAP_AMD_COARSE_MIN", "0")
    with eng.Index(idx, 0) as ix:
        forced, _ = ix.min_unique_segment(rec[:30_000_000 + kmax - 1], 30_000_000, kmin, kmax)
        assert np.array_equal(forced, whole[:30_000_000])


def test_human_shaped_full_size_oracle_windows(tmp_path, eng):
    """The human-shaped stand-in of configs[3]'s genome at FULL size (synth.human_like_dna, 3.09 Gbp in 24 records: repeat
    families at 2 - 20 % divergence on both strands, segmental duplications, soft-masked half, N runs; lower case is searched,
    newmap/search.py:23) on a device-built index -- what k_sweep and the routing of the open words were built for.  ORACLE
    contact by the scan counter, ONE pass over all 24 records for every claim of: range mode 20:200 on three windows (the
    start of chr1 behind its telomere gap, a repeat-rich stretch inside it, the end of chr21 with the tail rule) and list
    mode k = 36 (newmap/search.py:551-644: k where the 36-mer occurs once, 0 where it is repeated or holds an N) on two;
    beside it the sweep against k_resolve on 30 M positions, range and list."""
    from newmap_amd import synth
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    recs = synth.config_genome("hs")
    assert len(recs) == 24 and sum(r.size for _, r in recs) > 3_000_000_000
    fa = tmp_path / "hs.fa"
    synth.write_fasta(fa, recs)
    idx = tmp_path / "hs.awfmi"
    generate_fm_index(str(fa), str(idx), 8, 12, device=0)
    chr1, chr21 = recs[0][1].tobytes(), recs[20][1].tobytes()
    all_records = [r[1] for r in recs]
    kmin, kmax, W = 20, 200, 120_000
    first = next(i for i in range(0, len(chr1), 1000) if chr1[i:i + 1000].upper().count(b"N") == 0)    # behind the telomere gap
    with eng.Index(idx, 0) as ix:
        ix.set_segment_guard(False)                          # (pieces of records: the record check has its own tests)
        ix.set_sweep(True)                                   # from the first launch on (the default waits for the handle's first open positions)
        n = 30_000_000
        a0 = 60_000_000
        whole, _ = ix.min_unique_segment(chr1[a0:a0 + n + kmax - 1], n, kmin, kmax)
        assert ix.info()["last_range_kernel"] == 5
        ow = ix.open_words()
        assert sum(ow["positions"][:3]) > 5 * ow["positions"][3] > 0      # most open positions lie in dense words: the sweep's launch
        zeros = float(np.mean(whole == 0))
        assert 0.01 < zeros < 0.5 and ((whole == 0) | ((whole >= kmin) & (whole <= kmax))).all()
        listed, _ = ix.fixed_k_segment(chr1[a0:a0 + n + 35], n, [36])
        ix.set_sweep(False)                                  # every open position walks for itself (k_resolve): the same elements
        plain, _ = ix.min_unique_segment(chr1[a0:a0 + n + kmax - 1], n, kmin, kmax)
        assert np.array_equal(plain, whole)
        plain_l, _ = ix.fixed_k_segment(chr1[a0:a0 + n + 35], n, [36])
        assert np.array_equal(plain_l, listed)
        ix.set_sweep(True)
        head, _ = ix.min_unique_segment(chr1[first:first + W + kmax - 1], W, kmin, kmax)
        tail, _ = ix.min_unique_segment(chr21, len(chr21), kmin, kmax)
        # the densest stretch of open positions of the 30 M: where the sweep did most of its work
        dens = np.add.reduceat((whole > 40).astype(np.int64), np.arange(0, n - W, W))
        hot = int(np.argmax(dens)) * W
    # ---- the oracle's scan counter: every claim of the windows in one pass over the genome
    wins = [(chr1[first:first + W + kmax - 1], head), (chr1[a0 + hot:a0 + hot + W + kmax - 1], whole[hot:hot + W]), (chr21[-W:], tail[-W:])]
    blob, starts, lens, rels = [], [], [], []
    base = 0
    for seg, out in wins:
        st, ln, rel = rd.closed_form_claims(seg, out, kmin, kmax)
        blob.append(seg); starts.append(st + base); lens.append(ln); rels.append(rel)
        base += len(seg)
    for o in (hot, 0):                                       # list mode k = 36: unique 36-mer <=> 36, repeated <=> 0, an N inside <=> 0
        seg, out = chr1[a0 + o:a0 + o + W + 35], listed[o:o + W]
        amb = ~np.isin(np.frombuffer(seg, np.uint8), np.frombuffer(b"ACGTacgt", np.uint8))
        room_ok = np.convolve(amb.astype(np.int64), np.ones(36, np.int64))[35:35 + W] == 0          # no ambiguous byte in seg[p : p + 36]
        assert not out[~room_ok].any() and np.isin(out, (0, 36)).all()
        st = np.flatnonzero(room_ok)
        blob.append(seg); starts.append(st + base); lens.append(np.full(st.size, 36, np.int64)); rels.append((out[st] == 0).astype(np.int8))
        base += len(seg)
    starts, lens, rels = np.concatenate(starts), np.concatenate(lens), np.concatenate(rels)
    tot = rd.scan_total_counts(all_records, b"".join(blob), starts, lens, kmin)
    assert ((tot == 1) == (rels == 0)).all(), "a reported length is not the least unique one"
    assert (tot[rels == 1] >= 2).all() and starts.size > 4 * W


# ------------------------------------------------------------------ BASELINE configs, scaled down
def _records_fasta(recs):
    out = []
    for name, seq in recs:
        out.append(b">" + name.encode() + b"\n" + seq.tobytes() + b"\n")
    return b"".join(out)


def test_config3_human_shaped_24_records(tmp_path, eng):
    """BASELINE configs[2] at 1/1500 scale: 24 records, search-range 24:150, written through
    write_unique_counts with a batch that splits the larger records."""
    from newmap_amd import synth
    from newmap_amd.search import SearchConfig, write_unique_counts
    recs = synth.config_genome("c3", 2.0)
    fa, idx = _build_index(tmp_path, _records_fasta(recs), "c3")
    out = tmp_path / "out"
    out.mkdir()
    write_unique_counts(SearchConfig(fasta_filepaths=[fa], fmindex_filepaths=[idx],
                                     kmer_lengths=list(range(24, 151)), is_binary_search=True,
                                     kmer_batch_size=100_000, output_directory=out))
    oracle = rd.OracleIndex([s.tobytes() for _, s in recs])
    for i, (name, seq) in enumerate(recs):
        got = np.fromfile(out / f"{name}.unique.uint8", dtype=np.uint8)
        assert got.size == seq.size and not got[-23:].any() and ((got == 0) | (got >= 24)).all(), name
        if i % 4 == 0 or i == len(recs) - 1:           # (the oracle's closed form on every fourth record and the last: its time is the test's)
            want = rd.closed_form_min_unique(seq.tobytes(), oracle, 24, 150)
            assert np.array_equal(got, want), name
    eng.close_all()


def test_config5_tandem_repeats_20_255(tmp_path, eng):
    """BASELINE configs[4] at 1/1250 scale: 50 % tandem repeats, 20:255 (worst-case walk depth)."""
    from newmap_amd import synth
    recs = synth.config_genome("c5", 0.5)
    fa, idx = _build_index(tmp_path, _records_fasta(recs), "c5")
    rec = recs[0][1].tobytes()
    oracle = rd.OracleIndex([rec], max_depth=300)          # (as above: the full suffix order of a tandem-rich text is most of this test's time)
    want = rd.closed_form_min_unique(rec, oracle, 20, 255)
    with eng.Index(idx, 0) as ix:
        got, amb = ix.min_unique_segment(rec, len(rec), 20, 255)
        assert amb == 0 and np.array_equal(got, want)
        ix.set_kernel(1)
        got1, _ = ix.min_unique_segment(rec, len(rec), 20, 255)
        assert np.array_equal(got1, want)
    assert 0.2 < np.count_nonzero(want == 0) / want.size < 0.8      # repeats really are not unique


def test_repeat_probes_change_the_work_not_the_result(tmp_path, mixed_genome, eng):
    """k_repeat_probe: one probe per 64 positions settles the stretches that occur twice over more than
    kmax bases.  Same output with probes on and off (both kernels that consume them), most zero
    positions of a tandem-rich genome are settled, and the LF steps drop by an order of magnitude."""
    from newmap_amd import _lib, synth
    recs = synth.config_genome("c5", 2.0)
    fa, idx = _build_index(tmp_path, _records_fasta(recs), "c5p")
    rec = recs[0][1].tobytes()
    L = _lib.lib()
    d_seq, d_out, d_st = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    assert L.nm_dev_alloc(0, len(rec), ctypes.byref(d_seq)) == 0
    assert L.nm_dev_alloc(0, len(rec), ctypes.byref(d_out)) == 0
    assert L.nm_dev_alloc(0, 64, ctypes.byref(d_st)) == 0
    buf = np.frombuffer(rec, np.uint8)
    assert L.nm_dev_upload(0, d_seq, buf.ctypes.data, buf.size) == 0

    def run(ix):
        ix.min_unique_segment_dev(d_seq.value, len(rec), len(rec), 20, 255, True, 1, d_out.value, d_st.value)
        assert L.nm_dev_sync(0) == 0
        got, st = np.zeros(len(rec), np.uint8), np.zeros(8, np.uint64)
        assert L.nm_dev_download(0, got.ctypes.data, d_out, got.size) == 0
        assert L.nm_dev_download(0, st.ctypes.data, d_st, 64) == 0
        return got, st

    with eng.Index(idx, 0) as ix:
        assert ix.info()["repeat_probes"] == 1
        ix.set_count_steps(True)
        ix.set_sweep(True)                                  # (from the first launch on; the default waits until the handle has met open positions)
        for kernel in (0, 1):
            ix.set_kernel(kernel)
            ix.set_repeat_probes(True)
            on, st_on = run(ix)
            tally = ix.probe_tally()
            steps_on = int(st_on[3]) + tally["lf_steps"]
            ix.set_repeat_probes(False)
            off, st_off = run(ix)
            assert ix.probe_tally()["settled"] == 0
            assert np.array_equal(on, off), kernel
            assert int(st_on[0]) == int(st_off[0]) and int(st_on[7]) == int(st_off[7])
            zeros = int(np.count_nonzero(off == 0))
            assert tally["settled"] > 0.8 * zeros, (tally, zeros)
            if kernel == 0:
                # the sweep (k_sweep) takes the open positions of a long repeat at ONE step each, probes or not; one walk per
                # position (k_resolve, the A/B form) is what the probes save an order of magnitude on
                ix.set_sweep(False)
                off_walks, st_walks = run(ix)
                ix.set_sweep(True)
                assert np.array_equal(off_walks, off)
                assert int(st_off[3]) * 8 < int(st_walks[3]), (int(st_off[3]), int(st_walks[3]))
                st_off = st_walks
            assert steps_on * 8 < int(st_off[3]), (kernel, steps_on, int(st_off[3]))
    for d in (d_seq, d_out, d_st):
        L.nm_dev_free(0, d)
    g = mixed_genome                                               # N runs, soft-masked and reverse-complement copies
    with eng.Index(g["idx"], 0) as ix:
        for rec in (g["r1"], g["r2"]):
            for kmin, kmax in ((20, 200), (8, 30), (20, 1000), (3, 5)):
                ix.set_repeat_probes(True)
                a, amb_a = ix.min_unique_segment(rec, len(rec), kmin, kmax)
                ix.set_repeat_probes(False)
                b, amb_b = ix.min_unique_segment(rec, len(rec), kmin, kmax)
                assert np.array_equal(a, b) and amb_a == amb_b, (kmin, kmax)


@pytest.mark.parametrize("k", [36, 100])
def test_config4_fixed_k(tmp_path, eng, k):
    """BASELINE configs[3] shape (fixed-k list mode, k = 36 and 100) on a scaled human-shaped genome
    with N gaps; compared with the oracle's restatement of linear_search incl. the tail rule."""
    from newmap_amd import synth
    recs = synth.config_genome("c3", 1.0)[:6]
    recs = [(n, s.copy()) for n, s in recs]
    recs[0][1][5000:6000] = ord("N")
    recs[2][1][:300] = ord("N")
    fa, idx = _build_index(tmp_path, _records_fasta(recs), "c4")
    oracle = rd.OracleIndex([s.tobytes() for _, s in recs])
    with eng.Index(idx, 0) as ix:
        for name, seq in recs:
            data = seq.tobytes()
            seg = rd.Segment(name.encode(), data, True)
            want, _ = rd.linear_search_segment(oracle, seg, [k], k, np.uint8)
            got, _ = ix.fixed_k_segment(data, len(data), [k])
            assert np.array_equal(got, want), (name, k)


def test_config4_human_shaped_stand_in(tmp_path, eng):
    """BASELINE configs[3]'s genome is GRCh38 (on no box); its stand-in at 1/1000 scale: 24 records of synth.human_like_dna --
    interspersed 300-base and 6-kb repeat families on both strands, segmental duplications, half of the bases soft-masked,
    telomere / centromere / gap runs of N.  List mode k = 36 and k = 100 (and a list), the native driver's files, and
    range mode 20:200 against the oracle's restatement of the reference driver: lower-case bases are searched
    (newmap/search.py:23), a k-mer that holds an N drops its position (:593-596), the tail rule (:590)."""
    from newmap_amd import synth
    recs = synth.config_genome("hs", 3.0)
    assert len(recs) == 24
    blob = np.concatenate([s for _, s in recs])
    assert 0.4 < ((blob >= 97) & (blob <= 122)).mean() < 0.6 and 0.01 < (blob == ord("N")).mean() < 0.2
    fa, idx = _build_index(tmp_path, _records_fasta(recs), "hs")
    oracle = rd.OracleIndex([s.tobytes() for _, s in recs])
    with eng.Index(idx, 0) as ix:
        for name, seq in recs[:2] + recs[-1:]:              # (the oracle's time is the test's)
            data = seq.tobytes()
            seg = rd.Segment(name.encode(), data, True)
            for ks in ([36], [100], [24, 36, 50, 100]):
                want, _ = rd.linear_search_segment(oracle, seg, ks, max(ks), np.uint8)
                got, _ = ix.fixed_k_segment(data, len(data), ks)
                assert np.array_equal(got, want), (name, ks)
            got, amb = ix.min_unique_segment(data, len(data), 20, 200)
            assert amb == int(np.count_nonzero(seq == ord("N")))
            assert np.array_equal(got, rd.closed_form_min_unique(data, oracle, 20, 200)), name
        out = tmp_path / "hs_out"
        out.mkdir()
        before = ix.guard_segments()
        ix.search_fasta(fa, out, [36], False, True, 64 * 2000)
        assert ix.guard_segments() == before                # the FASTA is the indexed genome: no guard
        for name, seq in recs:
            data = seq.tobytes()
            want, _ = rd.linear_search_segment(oracle, rd.Segment(name.encode(), data, True), [36], 36, np.uint8)
            assert np.array_equal(np.fromfile(out / f"{name}.unique.uint8", dtype=np.uint8), want), name
        zeros = np.mean([np.mean(np.fromfile(out / f"{n_}.unique.uint8", dtype=np.uint8) == 0) for n_, _ in recs])
        assert 0.05 < zeros < 0.6                           # repeats and N runs leave a real share of positions without a unique 36-mer


def test_both_range_kernels_agree(mixed_genome, eng):
    """the two range paths -- one lane per position (k_min_unique) and the sites (k_sites + gated probes + k_resolve)
    -- are schedules of the same arithmetic: identical elements and ambiguity counts for every window / group size,
    with and without probes, on LF entries and on packed rank blocks, in the 32-bit and in the > 2^31-row
    instantiations, and both equal the oracle"""
    g = mixed_genome
    with eng.Index(g["idx"], 0) as ix:
        info = ix.info()
        assert 8 <= info["quad_small_core_length"] < info["quad_core_length"] <= info["seed_length"]
        assert max(info["seed_length"], info["quad_core_length"] + 4) < info["dict_length"] <= 24 and info["dict_entries"] > 0
        w = info["quad_small_core_length"] + 4                # the shortest window the sites can use
        for rec in (g["r1"], g["r2"]):
            for kmin, kmax in ((20, 200), (8, 30), (20, 1000), (w, 64), (w - 1, 64), (w + 1, 64), (60, 90), (61, 90), (62, 90),
                               (64, 64), (100, 300), (124, 200), (125, 200), (252, 255), (253, 255)):
                ix.set_force_big(False)
                ix.set_kernel(1)
                ix.set_repeat_probes(False)
                a, amb_a = ix.min_unique_segment(rec, len(rec), kmin, kmax)
                if (kmin, kmax) in ((20, 200), (w, 64), (252, 255)):
                    assert np.array_equal(a, rd.closed_form_min_unique(rec, g["oracle"], kmin, kmax))
                ix.set_lf2(False)                                 # single LF steps only (the two-base LF blocks are an A/B switch)
                a1, _ = ix.min_unique_segment(rec, len(rec), kmin, kmax)
                ix.set_lf2(True)
                assert info["lf2_blocks"] == 1 and np.array_equal(a, a1), ("two-base steps", kmin, kmax)
                sites_ok = w <= kmin <= 252                       # NM_SITE_MAX_KMIN
                for big in (False, True):
                    ix.set_force_big(big)
                    for kernel in (0, 1, 5):
                        ix.set_kernel(kernel)
                        for probes in (True, False):
                            ix.set_repeat_probes(probes)
                            for d_cap, table in (((59, 0), (0, 1), (2, 2), (59, 1), (59, 2)) if kernel == 5 and probes else ((59, 0),)):
                                ix.set_site_d(d_cap)
                                ix.set_site_table(table)          # picked per launch / long cores / short cores + second chance
                                c, amb_c = ix.min_unique_segment(rec, len(rec), kmin, kmax)
                                used = ix.info()["last_range_kernel"]
                                assert used == (5 if sites_ok and kernel != 1 else 1), (kernel, kmin, used)
                                assert np.array_equal(a, c) and amb_c == amb_a, (kernel, big, probes, d_cap, table, kmin, kmax)
                                if used == 5 and table == 2:
                                    assert ix.info()["last_site_core_length"] == info["quad_small_core_length"]
                                if used == 5 and table in (0, 2):   # the same without the repeat dictionary (second table + seed walks)
                                    ix.set_dictionary(False)
                                    c2, _ = ix.min_unique_segment(rec, len(rec), kmin, kmax)
                                    ix.set_dictionary(True)
                                    assert np.array_equal(a, c2), ("no dictionary", big, probes, d_cap, table, kmin, kmax)
                ix.set_site_d(59)
                ix.set_site_table(0)
                ix.set_force_big(False)
                ix.set_kernel(0)
                ix.set_repeat_probes(True)
                ix.set_lf_blocks(False)                           # packed 32-byte rank blocks instead of LF entries
                e, _ = ix.min_unique_segment(rec, len(rec), kmin, kmax)
                ix.set_lf_blocks(True)
                assert np.array_equal(a, e)
        # unaligned output pointer and 2-byte elements take the element-wise stores of k_sites
        rec = g["r1"][:70_001]
        ix.set_kernel(5)
        a16, _ = ix.min_unique_segment(rec, len(rec), 20, 300)
        ix.set_kernel(1)
        b16, _ = ix.min_unique_segment(rec, len(rec), 20, 300)
        assert a16.dtype == np.uint16 and np.array_equal(a16, b16)


def test_big_index_code_path(mixed_genome, eng):
    """every kernel templated for > 2^31 BWT rows (64-bit superblock table: the instantiations a 3 Gbp genome runs) on a
    small index, against the oracle: k_resolve / k_repeat_probe / k_repeat_probe_coarse / k_min_unique / k_fixed_k /
    k_count <true>, range and list mode, probes on and off, coarse probes forced"""
    g = mixed_genome
    with eng.Index(g["idx"], 0) as ix:
        rec = g["r1"]
        want = rd.closed_form_min_unique(rec, g["oracle"], 20, 200)
        want_n = rd.closed_form_min_unique(rec, g["oracle"], 20, 200, False)
        c = g["oracle"].count_from_sequence(rec, [0, 77, 50_000], [30, 12, 400])
        seg = rd.Segment(b"r", rec[:100_000], True)
        f36, _ = rd.linear_search_segment(g["oracle"], seg, [36], 36, np.uint8, True)
        f3, _ = rd.linear_search_segment(g["oracle"], seg, [24, 36, 100], 100, np.uint8, True)
        r_at = rec.find(b"R")
        assert r_at < 0 or r_at >= 100_000                   # (list mode and the lone R: test_list_mode_iupac_divergence)
        ix.set_force_big(True)
        for kernel in (0, 1, 5):
            ix.set_kernel(kernel)
            for probes in (True, False):
                ix.set_repeat_probes(probes)
                b, amb_b = ix.min_unique_segment(rec, len(rec), 20, 200)
                assert np.array_equal(want, b) and amb_b == sum(ch not in b"ACGTacgt" for ch in rec), (kernel, probes)
                assert np.array_equal(f36, ix.fixed_k_segment(rec[:100_000], 100_000, [36])[0]), (kernel, probes)
                assert np.array_equal(f3, ix.fixed_k_segment(rec[:100_000], 100_000, [24, 36, 100])[0]), (kernel, probes)
        ix.set_kernel(0)
        ix.set_repeat_probes(True)
        assert np.array_equal(c, ix.count_from_sequence(rec, [0, 77, 50_000], [30, 12, 400]))
        assert np.array_equal(want_n, ix.min_unique_segment(rec, len(rec), 20, 200, use_revcomp=False)[0])


def test_coarse_probes_forced_big_and_small(mixed_genome, eng, monkeypatch):
    """k_repeat_probe_coarse in both instantiations (forced on: NEWMAP_AMD_COARSE=2 with a launch threshold of 0), on the
    genome with the 30 kb tandem array, against the oracle"""
    g = mixed_genome
    monkeypatch.setenv("NEWMAP_AMD_COARSE", "2")
    monkeypatch.setenv("NEWMAP_AMD_COARSE_MIN", "0")
    rec = g["r1"]
    want = rd.closed_form_min_unique(rec, g["oracle"], 20, 100)
    with eng.Index(g["idx"], 0) as ix:
        ix.set_count_steps(True)
        ix.set_sweep(True)                                   # (probes from the first launch on)
        for big in (False, True):
            ix.set_force_big(big)
            for kernel in (0, 1):
                ix.set_kernel(kernel)
                got, _ = ix.min_unique_segment(rec, len(rec), 20, 100)
                assert np.array_equal(got, want), (big, kernel)
                assert ix.probe_tally()["settled"] > 20_000     # the tandem array is settled by the probes


def test_device_built_index_searched_against_oracle(tmp_path, mixed_genome, eng):
    """nm_index_build_device (suffix sort on the GPU, both the 32-bit and the bucketed 64-bit variant): the index it
    writes is SEARCHED against the oracle, not only compared with the host builder's file"""
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    g = mixed_genome
    want = {kk: rd.closed_form_min_unique(g["r1"], g["oracle"], *kk) for kk in ((20, 200), (24, 150))}
    for variant in ("", "large"):
        if variant:
            os.environ["NEWMAP_AMD_DEVICE_SA"] = variant
        try:
            idx = tmp_path / f"dev{variant}.awfmi"
            generate_fm_index(str(g["fa"]), str(idx), 8, 12, device=0)
        finally:
            os.environ.pop("NEWMAP_AMD_DEVICE_SA", None)
        with eng.Index(idx, 0) as ix:
            for kk, w in want.items():
                got, _ = ix.min_unique_segment(g["r1"], len(g["r1"]), *kk)
                assert np.array_equal(got, w), (variant, kk)
            starts = np.arange(0, 60_000, 997)
            lens = (starts % 90) + 1
            assert np.array_equal(ix.count_from_sequence(g["r2"], starts, lens),
                                  g["oracle"].count_from_sequence(g["r2"], starts, lens))


def test_list_mode_iupac_divergence(mixed_genome, eng):
    """DOCUMENTED DIVERGENCE: list mode in the reference drops a position only when its k-mer
    contains an upper-case N (newmap/search.py:593); any other non-ACGT byte stays in the query and is counted by
    AwFmIndex's ambiguity letter (unpinned: the library is absent).  The engine drops a position on ANY non-ACGT byte
    inside its k-mer.  r1 holds one lone 'R': the two differ exactly on the k positions whose k-mer covers it -- there
    the oracle's restatement of the reference (R matches R literally, so the k-mer is found once) reports k, the
    engine 0 -- and nowhere else."""
    g = mixed_genome
    rec = g["r1"]
    r_at = rec.find(b"R")
    assert r_at == 250_000
    k = 36
    lo, hi = r_at - 2000, r_at + 2000
    piece = rec[lo:hi]
    seg = rd.Segment(b"r", piece, True)
    want, _ = rd.linear_search_segment(g["oracle"], seg, [k], k, np.uint8, True)
    with eng.Index(g["idx"], 0) as ix:
        got, _ = ix.fixed_k_segment(piece, len(piece), [k])
    differ = np.flatnonzero(got != want)
    covers = np.arange(r_at - lo - k + 1, r_at - lo + 1)
    assert set(differ.tolist()) <= set(covers.tolist())
    assert (got[covers] == 0).all()
    # (the restated reference: positions covering the R but not starting on it may be reported k -- the engine never does)
    assert differ.size == int(np.count_nonzero(want[covers] != 0))


def test_native_driver_front_ends_and_shards(tmp_path, mixed_genome, eng, monkeypatch):
    """csrc/nm_driver.hip: the parallel front-end (mapped file, threaded strip, slot ring, writer pool) == the streaming
    one (forced) on an odd FASTA and on the mixed genome; and the shares of a 3-rank job (nm_search_fasta_shard, called
    rank by rank here; small interleaved chunks) add up to the single-process files, for range and list mode."""
    rng = np.random.default_rng(78)
    body = _random_dna(rng, 9000)
    odd = (b"ACGTACGTTTGACCA" + body[:200] + b"\n>r1 first\n" + body[200:1500] + b"\n" + body[1500:1700] + b"  \n"
           b">r1 again\n" + body[1700:2500] + b"\n;r2\r\n" + body[2500:3300] + b"\r\n>empty\n>r3\n" + body[3300:6000] +
           b"\n>r1\n" + body[6000:6100] + b"\n\n>r4 x\n" + b"\n".join(body[6100 + i:6100 + i + 61] for i in range(0, 2900, 61)) + b"\nNNNN")
    fa_odd, idx_odd = _build_index(tmp_path, odd, "odd")
    g = mixed_genome

    def run(ix, fa, out, ks, is_range, batch, rank=0, world=1):
        out.mkdir(exist_ok=True)
        return ix.search_fasta(fa, out, ks, is_range, True, batch, rank=rank, world=world)

    def files(d):
        return {p.name: p.read_bytes() for p in sorted(d.iterdir())}

    for fa, idx, tag in ((fa_odd, idx_odd, "odd"), (g["fa"], g["idx"], "mixed")):
        with eng.Index(idx, 0) as ix:
            for ks, is_range, batch in (([12, 60], True, 700), ([20, 200], True, 100_000), ([24], False, 50_000), ([30, 16, 40], False, 1300)):
                name = f"{tag}_{ks[0]}_{int(is_range)}_{batch}"
                monkeypatch.setenv("NEWMAP_AMD_STREAMING_DRIVER", "1")
                t_stream = run(ix, fa, tmp_path / (name + "_s"), ks, is_range, batch)
                monkeypatch.setenv("NEWMAP_AMD_STREAMING_DRIVER", "0")
                t_fast = run(ix, fa, tmp_path / (name + "_f"), ks, is_range, batch)
                a, b = files(tmp_path / (name + "_s")), files(tmp_path / (name + "_f"))
                assert a == b and a, name
                for k in ("positions", "ambiguous", "unique", "no_unique", "max_len", "min_len"):
                    # (the streaming reader also counts the earlier run of an id that comes back: "r1" in the odd file)
                    assert tag == "odd" or t_stream[k] == t_fast[k], (name, k)
                monkeypatch.setenv("NEWMAP_AMD_SHARD_CHUNK", "997")
                parts = [run(ix, fa, tmp_path / (name + "_w3"), ks, is_range, batch, rank=r, world=3) for r in range(3)]
                monkeypatch.delenv("NEWMAP_AMD_SHARD_CHUNK")
                assert files(tmp_path / (name + "_w3")) == b, name
                assert sum(p["positions"] for p in parts) == t_fast["positions"] and sum(p["unique"] for p in parts) == t_fast["unique"]
    # gzip input (newmap/util.py:10-18) through the same front-ends: inflated once, then stripped in parallel; one process and
    # the shares of three ranks give the plain file's bytes
    import gzip
    for fa, idx, tag in ((fa_odd, idx_odd, "odd"), (g["fa"], g["idx"], "mixed")):
        gz = tmp_path / f"{tag}.fa.gz"
        gz.write_bytes(gzip.compress(Path(fa).read_bytes(), 1))
        with eng.Index(idx, 0) as ix:
            for ks, is_range, batch in (([20, 200], True, 100_000), ([30, 16, 40], False, 1300)):
                name = f"{tag}_gz_{ks[0]}_{int(is_range)}"
                plain = run(ix, fa, tmp_path / (name + "_p"), ks, is_range, batch)
                one = run(ix, gz, tmp_path / (name + "_g"), ks, is_range, batch)
                assert files(tmp_path / (name + "_g")) == files(tmp_path / (name + "_p")) and one["positions"] == plain["positions"], name
                monkeypatch.setenv("NEWMAP_AMD_SHARD_CHUNK", "1999")
                parts = [run(ix, gz, tmp_path / (name + "_g3"), ks, is_range, batch, rank=r, world=3) for r in range(3)]
                monkeypatch.delenv("NEWMAP_AMD_SHARD_CHUNK")
                assert files(tmp_path / (name + "_g3")) == files(tmp_path / (name + "_p")), name
                assert sum(p_["positions"] for p_ in parts) == plain["positions"]
    with eng.Index(idx_odd, 0) as ix:                       # include / exclude and a mismatching FASTA through the parallel front-end
        t = run(ix, fa_odd, tmp_path / "inc", [12, 60], True, 700)
        assert t["records"] == 5                            # "", r1 (its last run), r2, r3, r4
        out = tmp_path / "inc2"
        out.mkdir()
        assert ix.search_fasta(fa_odd, out, [12, 60], True, True, 700, include=[b"r3"])["records"] == 1
        assert sorted(p.name for p in out.iterdir()) == ["r3.unique.uint8"]
        other = tmp_path / "other.fa"
        other.write_bytes(b">zz\n" + _random_dna(rng, 3000) + b"\n")
        with pytest.raises(RuntimeError, match="The following generated k-mer was not found in the index:\n[ACGT]+\n"):
            run(ix, other, tmp_path / "bad", [12, 60], True, 700)


def test_native_driver_equals_python_driver(tmp_path, golden_search, eng, monkeypatch):
    """csrc/nm_driver.hip (streaming FASTA reader + pipelined launches) writes the same files as the
    segment-by-segment Python loop and as the reference fixtures; also .gz input, include / exclude,
    consecutive records with one id, headerless data."""
    import gzip
    from newmap_amd.search import SearchConfig, write_unique_counts

    def run(fa, idx, out, python_driver, **kw):
        out.mkdir()
        monkeypatch.setenv("NEWMAP_AMD_PYTHON_DRIVER", "1" if python_driver else "0")
        write_unique_counts(SearchConfig(fasta_filepaths=[fa], fmindex_filepaths=[idx], output_directory=out, **kw))
        return {p.name: p.read_bytes() for p in sorted(out.iterdir())}

    for i, c in enumerate(golden_search):
        d = tmp_path / f"n{i}"
        d.mkdir()
        fa, idx = _build_index(d, c["fasta"].encode("latin-1"))
        kw = dict(kmer_lengths=c["kmer_lengths"], is_binary_search=c["is_binary"], kmer_batch_size=c["batch"],
                  use_reverse_complement=c["use_reverse_complement"])
        a = run(fa, idx, d / "native", False, **kw)
        b = run(fa, idx, d / "python", True, **kw)
        assert a == b, c["name"]
        if "quirk" in c["name"]:                           # (the documented divergence: asserted in the fixture test above)
            continue
        for rid, e in c["expected"].items():
            assert np.frombuffer(a[f"{rid}.unique.{e['dtype']}"], dtype=e["dtype"]).tolist() == e["values"]
    # odd FASTA shapes
    rng = np.random.default_rng(77)
    body = _random_dna(rng, 5000)
    text = (b"ACGTACGTTTGACCA" + body[:200] + b"\n>r1 first\n" + body[200:1500] + b"\n" + body[1500:1700] + b"  \n"
            b">r1 again\n" + body[1700:2500] + b"\n;r2\r\n" + body[2500:3300] + b"\r\n>empty\n>r3\n" + body[3300:] + b"\nNNNN")
    d = tmp_path / "odd"
    d.mkdir()
    fa, idx = _build_index(d, text)
    gz = d / "x.fa.gz"
    with gzip.open(gz, "wb") as fh:
        fh.write(text)
    kw = dict(kmer_lengths=list(range(12, 61)), is_binary_search=True, kmer_batch_size=300)
    a = run(fa, idx, d / "native", False, **kw)
    b = run(fa, idx, d / "python", True, **kw)
    g = run(gz, idx, d / "gz", False, **kw)
    assert a == b == g and set(a) == {".unique.uint8", "r1.unique.uint8", "r2.unique.uint8", "r3.unique.uint8"}
    assert len(a["r1.unique.uint8"]) == 1300 + 200 + 800          # consecutive records with one id append
    inc = run(fa, idx, d / "inc", False, include_sequence_ids=[b"r2", b"r3"], **kw)
    assert set(inc) == {"r2.unique.uint8", "r3.unique.uint8"} and inc["r2.unique.uint8"] == a["r2.unique.uint8"]
    exc = run(fa, idx, d / "exc", False, exclude_sequence_ids=[b"r1", b""], **kw)
    assert exc == inc
    with pytest.raises(ValueError, match="None of the included sequences"):
        run(fa, idx, d / "none", False, include_sequence_ids=[b"zzz"], **kw)
    eng.close_all()


def test_track_on_device_matches_reference(tmp_path, eng):
    """csrc/nm_track.hip: BED + WIG bytes equal the reference's newmap/track.py (fixtures written by the
    reference itself) and the host numpy path, incl. a multi-tile input for the hierarchical scans"""
    import json
    from newmap_amd import track
    golden = json.loads((Path(__file__).resolve().parent / "golden" / "golden_track.json").read_text())
    files = []
    for name, a in golden["arrays"].items():
        p = tmp_path / f"{name}.unique.{a['dtype']}"
        np.array(a["values"], dtype=a["dtype"]).tofile(p)
        files.append(p)
    for c in golden["cases"]:
        bed, wig = tmp_path / "d.bed", tmp_path / "d.wig"
        track.write_mappability_files(files, c["k"], str(bed), str(wig), False)
        assert bed.read_text() == c["bed"], c["k"]
        assert wig.read_text() == c["wig"], c["k"]
    # larger than one scan tile, three levels deep for the flag scan
    rng = np.random.default_rng(8)
    big = np.where(rng.random(20_000_000) < 0.4, 0, rng.integers(20, 120, 20_000_000)).astype(np.uint8)
    big[5_000_000:5_300_000] = 0
    p = tmp_path / "big.unique.uint8"
    big.tofile(p)
    for k in (24, 100):
        out = {}
        for mode in ("device", "host"):
            os.environ["NEWMAP_AMD_TRACK"] = mode
            bed, wig = tmp_path / f"{mode}.bed", tmp_path / f"{mode}.wig"
            track.write_mappability_files([p], k, str(bed), str(wig), False)
            out[mode] = (bed.read_bytes(), wig.read_bytes())
        os.environ.pop("NEWMAP_AMD_TRACK")
        assert out["device"][0] == out["host"][0]
        assert out["device"][1] == out["host"][1]


def test_multi_fasta_multi_index_mode(tmp_path, eng):
    """several FASTA files in lock-step x several indexes through write_unique_counts,
    against fixtures written by the reference driver in that mode (tests/golden/make_golden_multi.py)"""
    import json
    from newmap_amd.search import SearchConfig, write_unique_counts
    cases = json.loads((Path(__file__).resolve().parent / "golden" / "golden_multi.json").read_text())["cases"]
    sharded_cases = 0
    for n, c in enumerate(cases):
        d = tmp_path / f"m{n}"
        d.mkdir()
        fas, idxs = [], []
        for i, t in enumerate(c["fastas"]):
            fa, idx = _build_index(d, t.encode("latin-1"), f"g{i}")
            fas.append(fa)
            idxs.append(idx)
        out = d / "out"
        out.mkdir()
        write_unique_counts(SearchConfig(fasta_filepaths=fas, fmindex_filepaths=idxs, kmer_lengths=c["kmer_lengths"],
                                         is_binary_search=c["is_binary"], kmer_batch_size=c["batch"],
                                         output_directory=out, use_reverse_complement=c["use_reverse_complement"]))
        for rid, e in c["expected"].items():
            got = np.fromfile(out / f"{rid}.unique.{e['dtype']}", dtype=e["dtype"])
            assert got.tolist() == e["values"], (c["name"], rid)
        # the sharded entry point in this mode (one rank here; the plan over ranks is tests/test_parallel_gloo.py): same files
        from newmap_amd.parallel import write_unique_counts_distributed
        out2 = d / "out_sharded"
        out2.mkdir()
        try:
            write_unique_counts_distributed(SearchConfig(fasta_filepaths=fas, fmindex_filepaths=idxs, kmer_lengths=c["kmer_lengths"],
                                                         is_binary_search=c["is_binary"], kmer_batch_size=c["batch"],
                                                         output_directory=out2, use_reverse_complement=c["use_reverse_complement"]))
        except ValueError as e:                    # files whose records differ in length are refused there, by name
            assert "record lengths" in str(e), e
            continue
        sharded_cases += 1
        for rid, e in c["expected"].items():
            assert (out2 / f"{rid}.unique.{e['dtype']}").read_bytes() == (out / f"{rid}.unique.{e['dtype']}").read_bytes(), (c["name"], rid)
    assert sharded_cases > 0
    eng.close_all()


def test_device_index_builder_writes_the_same_file(tmp_path, eng, monkeypatch):
    """suffix array by prefix doubling on the GPU (rocPRIM radix sorts) -> index file
    byte-identical to the host builder's, on uniform, repeat-heavy and multi-record inputs"""
    from newmap_amd import synth
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    rng = np.random.default_rng(17)
    inputs = {
        "uniform": b">u\n" + synth.config_genome("c2", 2.0)[0][1].tobytes() + b"\n",
        "tandem": b">t\n" + synth.tandem_dna(1_000_000, 5).tobytes() + b"\n",
        "mixed": (b">a x\n" + _random_dna(rng, 300_000) + b"NNNN" + b"A" * 50_000 + b"\n>b\n" + b"ACG" * 40_000 +
                  _random_dna(rng, 100_000).lower() + b"\n>c\nT\n"),
        "tiny": b">x\nACGT\n",
    }
    for name, text in inputs.items():
        fa = tmp_path / f"{name}.fa"
        fa.write_bytes(text)
        host, dev, big = tmp_path / f"{name}.host.awfmi", tmp_path / f"{name}.dev.awfmi", tmp_path / f"{name}.big.awfmi"
        generate_fm_index(str(fa), str(host), 8, 12)
        generate_fm_index(str(fa), str(dev), 8, 12, device=0)
        assert host.read_bytes() == dev.read_bytes(), name
        # the path for texts of 2^31 symbols and more (64-bit, bucketed initial sort, refinement of the tied
        # groups only, BWT gathered on the device), forced onto the small text
        monkeypatch.setenv("NEWMAP_AMD_DEVICE_SA", "large")
        generate_fm_index(str(fa), str(big), 8, 12, device=0)
        monkeypatch.delenv("NEWMAP_AMD_DEVICE_SA")
        assert host.read_bytes() == big.read_bytes(), name


def test_cli_end_to_end(tmp_path):
    """reference tests/test_end_to_end.sh, with assertions: `index`, `search`, `track` through the CLI"""
    import json
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    golden = root / "tests" / "golden"
    env = dict(os.environ, PYTHONPATH=str(root))

    def cli(*args):
        return subprocess.run([sys.executable, "-m", "newmap_amd.main", *args], cwd=tmp_path, env=env, check=True,
                              capture_output=True)

    cli("index", "--compression-ratio=32", "--seed-length=1", str(golden / "genome.fa"))
    assert (tmp_path / "genome.awfmi").read_bytes()[:8] == b"NMAPGFX1"          # index.py:12-15 default name
    cli("search", "--search-range=4:10", "--kmer-batch-size=15", "--verbose", str(golden / "genome.fa"), "genome.awfmi")
    chr1 = np.fromfile(tmp_path / "chr1.unique.uint8", dtype=np.uint8).tolist()
    chr2 = np.fromfile(tmp_path / "chr2.unique.uint8", dtype=np.uint8).tolist()
    assert chr1 == [0, 10, 9, 8, 7, 6, 5, 4, 4, 4, 6, 5, 4, 4, 4, 0, 0, 0, 0, 0]       # tests/test_unique_counts.py:17-21
    assert chr2 == [10, 10, 9, 8, 7, 6, 5, 4, 4, 4] + [0] * 20
    cli("track", "--multi-read", "genome.10.wig", "--single-read", "genome.10.bed", "10",
        "chr1.unique.uint8", "chr2.unique.uint8")
    want = [c for c in json.loads((golden / "golden_track.json").read_text())["cases"] if c["k"] == 10][0]
    got_bed = (tmp_path / "genome.10.bed").read_text().splitlines()
    want_bed = [l for l in want["bed"].splitlines() if l.startswith(("chr1\t", "chr2\t"))]
    assert got_bed == want_bed
    bed_stdout = cli("track", "chr1.unique.uint8").stdout.decode()              # k defaults to 24, BED to stdout
    assert bed_stdout.startswith("chr1\t0\t")
    # list mode and a fixed k through the CLI
    cli("search", "--search-range=4,6,10", "-o", "lin", str(golden / "genome.fa"), "genome.awfmi")
    assert (tmp_path / "lin" / "chr1.unique.uint8").stat().st_size == 20


def test_randomised_soak_fast_paths_against_plain_kernels(monkeypatch):
    """tools/fuzz_gpu.py, a few rounds: quad table + repeat probes + list-mode routing against the plain
    one-lane-per-position kernels on random genomes with tandem arrays, copies on both strands, ambiguity runs and
    soft-masked stretches, cut at random batch sizes (the long runs are recorded in DESIGN.md)."""
    import sys
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "tools"))
    import fuzz_gpu
    monkeypatch.setattr(sys, "argv", ["fuzz_gpu.py", "--rounds", "4", "--seed", "77"])
    fuzz_gpu.main()

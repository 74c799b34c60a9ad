"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/newmap_amd.h declares (no compute calls), CLI / SearchConfig parsing follows the
reference's rules, the FASTA front-end reproduces the reference's segments, and the engine
refuses to run without a device."""
import argparse
import ctypes
import io
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    from newmap_amd import _lib
    header = (ROOT / "include" / "newmap_amd.h").read_text()
    declared = set(re.findall(r"\b(nm_[a-z_0-9]+)\s*\(", header))
    declared -= {"nm_index"}
    assert declared, "no declarations found"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = ctypes.CDLL(str(_lib.LIB_PATH))
    for sym in declared:
        assert hasattr(L, sym), sym
    assert b"gfx950" in _lib.lib().nm_version()


def test_no_cpu_path(tmp_path):
    """device < 0 is refused; without a GPU nm_index_open fails loudly instead of falling back."""
    from newmap_amd._c_newmap_generate_index import generate_fm_index
    from newmap_amd.engine import Index, device_count
    fa = tmp_path / "x.fa"
    fa.write_bytes(b">x\nACGTACGTAC\n")
    idx = tmp_path / "x.awfmi"
    generate_fm_index(str(fa), str(idx), 8, 12)
    with pytest.raises(RuntimeError, match="no CPU path"):
        Index(idx, -1)
    if device_count() == 0:
        with pytest.raises(RuntimeError):
            Index(idx, 0)


def test_product_never_imports_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b)|(#include\s+[\"<][^\n]*oracle)|liboracle|hostsim", re.MULTILINE)
    for p in (ROOT / "newmap_amd").rglob("*"):
        if p.suffix in (".py", ".h", ".hpp", ".cpp", ".hip"):
            hits = [m.group(0) for m in bad.finditer(p.read_text())]
            # nm_core.h and nm_fasta_scan.hpp name the host simulator in a comment only
            hits = [h for h in hits if not (p.name in ("nm_core.h", "nm_fasta_scan.hpp") and h == "hostsim")]
            assert not hits, (p, hits)


def _args(**kw):
    base = dict(fasta_file="g.fa", index_file=None, search_range="20:200", output_directory=".",
                initial_search_length=0, include_sequences=None, exclude_sequences=None, norc=False,
                verbose=False, num_threads=1, kmer_batch_size=10_000_000, device=None)
    base.update(kw)
    return argparse.Namespace(**base)


def test_search_config_from_args(tmp_path, monkeypatch):
    from newmap_amd.search import SearchConfig
    monkeypatch.chdir(tmp_path)
    (tmp_path / "g.awfmi").write_bytes(b"x")
    c = SearchConfig.from_args(_args())
    assert c.is_binary_search and c.kmer_lengths[0] == 20 and c.kmer_lengths[-1] == 200
    assert c.fmindex_filepaths == [Path("g.awfmi")] and c.use_reverse_complement
    c = SearchConfig.from_args(_args(search_range="36,100", norc=True, include_sequences="chr1,chr2"))
    assert not c.is_binary_search and c.kmer_lengths == [36, 100] and not c.use_reverse_complement
    assert c.include_sequence_ids == [b"chr1", b"chr2"]
    with pytest.raises(ValueError, match="range start length is larger"):
        SearchConfig.from_args(_args(search_range="30:20"))
    with pytest.raises(ValueError, match="Could not parse"):
        SearchConfig.from_args(_args(search_range="a:b"))
    with pytest.raises(ValueError, match="Initial search length"):
        SearchConfig.from_args(_args(search_range="20,30", initial_search_length=5))
    with pytest.raises(ValueError, match="both include and exclude"):
        SearchConfig.from_args(_args(include_sequences="a", exclude_sequences="b"))
    with pytest.raises(FileNotFoundError):
        SearchConfig.from_args(_args(index_file="missing.awfmi"))


def test_cli_flags_match_the_reference():
    from newmap_amd.main import build_parser
    p = build_parser()
    a = p.parse_args(["index", "genome.fa", "--seed-length", "1", "--compression-ratio", "1", "-i", "o.awfmi"])
    assert (a.seed_length, a.compression_ratio, a.output) == (1, 1, "o.awfmi")
    a = p.parse_args(["search", "genome.fa", "idx.awfmi", "--search-range=24:150", "-o", "out", "-t", "20",
                      "-s", "5000", "--norc", "-l", "30", "-i", "chr1"])
    assert a.search_range == "24:150" and a.num_threads == 20 and a.kmer_batch_size == 5000 and a.norc
    a = p.parse_args(["search", "genome.fa"])
    assert a.search_range == "20:200" and a.index_file is None and a.kmer_batch_size == 10_000_000
    a = p.parse_args(["track", "24", "chr1.unique.uint8", "-m", "-"])
    assert a.read_length == "24" and a.multi_read == "-"


def test_output_type_and_num_kmers():
    from newmap_amd.fasta import SequenceSegment
    from newmap_amd.search import get_num_kmers, output_type
    assert output_type(255)[1] == "uint8" and output_type(256)[1] == "uint16" and output_type(65536)[1] == "uint32"
    assert get_num_kmers(SequenceSegment(b"x", b"A" * 30, False), 10) == 21
    assert get_num_kmers(SequenceSegment(b"x", b"A" * 30, True), 10) == 30


def test_fasta_front_end_matches_reference_segments(golden_host):
    from newmap_amd.fasta import sequence_segments
    for c in golden_host["segments"]:
        got = [[s.id.decode("latin-1"), s.data.decode("latin-1"), s.epilogue]
               for s in sequence_segments(io.BytesIO(c["text"].encode("latin-1")), c["length"], c["overlap"])]
        assert got == c["expected"], (c["name"], c["length"], c["overlap"])


def test_fasta_front_end_odd_whitespace_and_blocks(monkeypatch):
    """slow path (inner blanks, trailing tabs, lone CR) and block boundaries agree with a plain
    restatement of newmap/fasta.py:47-79"""
    from newmap_amd import fasta
    from oracle import ref_driver as rd
    text = b"AC GT \nNN\t\n>r1 d\nACGT\rAC\r\nGG  \n;r2\n\n \nTT\n>r3\n>r4\nA" + b"\nACGTACGTAC" * 50
    want = rd.read_records(io.BytesIO(text).readlines())
    for block in (7, 16, 1 << 20):
        monkeypatch.setattr(fasta, "_BLOCK_BYTES", block)
        assert list(fasta.fasta_records(io.BytesIO(text))) == want, block


def test_synthetic_generators_are_seeded():
    from newmap_amd import synth
    a = synth.config_genome("c2", 0.01)[0][1]
    b = synth.config_genome("c2", 0.01)[0][1]
    assert np.array_equal(a, b) and a.size == 10_000 and set(np.unique(a)) <= set(b"ACGT")
    t = synth.tandem_dna(200_000, 20260517)
    assert t.size == 200_000
    recs = synth.config_genome("c3", 1.0)
    assert len(recs) == 24 and recs[0][0] == "chr1" and recs[-1][0] == "chrY"


def test_native_fasta_scan_equals_python_reader(tmp_path, golden_host):
    """csrc/nm_fasta_scan.hpp -- the threaded FASTA scan of the native driver's parallel front-end -- yields the records
    of newmap_amd.fasta.fasta_records (itself pinned by the reference-generated segment fixtures): ids, data with every
    line stripped of trailing whitespace, ';' headers, data in front of any header, records without data, CR LF, a
    last line without newline; for pieces of a few bytes and of megabytes, 1 and 5 threads; and arbitrary ranges of a
    record read back on their own (what a rank of a sharded job does)."""
    import io
    import numpy as np
    from newmap_amd.fasta import fasta_records
    from tests import hostsim
    rng = np.random.default_rng(3)
    body = bytes(np.frombuffer(b"ACGTNacgtn", np.uint8)[rng.integers(0, 10, 40_000)])
    texts = [
        b"", b"\n\n", b">only header\n", b"ACGT", b"ACGT\n", b">a\nAC GT \t\n>b\n\n>c desc here\nAAA\r\nCC\r\n",
        b"ACGTACGTTTGACCA" + body[:200] + b"\n>r1 first\n" + body[200:1500] + b"\n" + body[1500:1700] + b"  \n"
        b">r1 again\n" + body[1700:2500] + b"\n;r2\r\n" + body[2500:3300] + b"\r\n>empty\n>r3\n" + body[3300:6000] +
        b"\n>r1\n" + body[6000:6100] + b"\n\n>r4 x\n" + b"\n".join(body[6100 + i:6100 + i + 61] for i in range(0, 2900, 61)) + b"\nNNNN",
        b">big one\n" + b"\n".join(body[i:i + 70] for i in range(0, len(body), 70)) + b"\n>x>y;z\n" + body[:999] + b" \n \n" + body[999:1500],
        b";c\n>\nAC\n> spaced id\nGG\n>\tt\nTT\n",
    ]
    for c in golden_host.get("segments", [])[:20]:
        if "fasta" in c:
            texts.append(c["fasta"].encode("latin-1"))
    for k, text in enumerate(texts):
        fa = tmp_path / f"f{k}.fa"
        fa.write_bytes(text)
        want = list(fasta_records(io.BytesIO(text)))
        for threads, chunk in ((1, 4 << 20), (5, 7), (3, 64), (4, 1000)):
            got = hostsim.fasta_scan(fa, threads, chunk)
            assert got == want, (k, threads, chunk)
    # ranges of a record, read back alone
    fa = tmp_path / "f7.fa"
    want = list(fasta_records(io.BytesIO(texts[7])))
    data = want[0][1]
    ranges = [(0, 0, 1), (0, 5, 5000), (0, len(data) - 1, len(data)), (0, 12345, 12346), (0, 0, len(data))] + \
             [(0, int(a), int(a) + int(b)) for a, b in zip(rng.integers(0, len(data) - 3000, 20), rng.integers(1, 3000, 20))]
    for threads, chunk in ((1, 50), (4, 333), (2, 1 << 20)):
        _, lens, pieces = hostsim.fasta_scan(fa, threads, chunk, ranges)
        assert lens[0] == len(data)
        for (i, lo, hi), piece in zip(ranges, pieces):
            assert piece == data[lo:hi], (threads, chunk, lo, hi)


def test_cli_imports_stay_light():
    """the one-shot CLI's native path needs no numpy on the Python side: importing the command line must not import it
    (0.2 s of a 0.9 s process on a 3 Gbp genome); the first array use does"""
    import subprocess
    import sys
    code = ("import sys; import newmap_amd.main; assert 'numpy' not in sys.modules, 'numpy imported by the CLI modules'; "
            "from newmap_amd import engine; assert engine.np.dtype('uint8').itemsize == 1 and 'numpy' in sys.modules; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_synthetic_genomes_do_not_depend_on_the_thread_pool():
    """the records of the bench genomes come from their own seeds: generated by a thread pool, same bytes as one by one"""
    from newmap_amd import synth
    got = synth.config_genome("c3", 0.5)
    f = 0.5e6 / sum(synth.HUMAN_SHAPED)
    assert len(got) == 24 and got[0][0] == "chr1" and got[-1][0] == "chrY"
    for i, (name, seq) in enumerate(got):
        assert np.array_equal(seq, synth.uniform_dna(max(1000, int(synth.HUMAN_SHAPED[i] * f)), 20260516 + i)), name
    hs = synth.config_genome("hs", 3.0)
    for i in (0, 7, 23):
        assert np.array_equal(hs[i][1], synth.human_like_dna(max(100_000, int(synth.HUMAN_SHAPED[i] * 3.0e6 / sum(synth.HUMAN_SHAPED))), 20260600 + i))

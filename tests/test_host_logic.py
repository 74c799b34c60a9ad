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
            # nm_core.h names the host simulator in a comment only
            hits = [h for h in hits if not (p.name == "nm_core.h" and h == "hostsim")]
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

"""`newmap track`: unique-length arrays -> single-read BED / multi-read WIG mappability tracks
(reference: newmap/track.py).  Downstream consumer of the `search` output (SURVEY.md section 8(f)
rank 2).  `write_mappability_files` runs on the device (csrc/nm_track.hip: scans, stream compaction,
text scatter) and fails loudly when no GPU is visible.  The numpy functions below are the same
arithmetic as a separate HOST tool, used only when asked for explicitly (NEWMAP_AMD_TRACK=host) -- never
as a silent fallback.  Both write the reference's bytes (tests/test_track.py, tests/test_gpu_parity.py).
"""
from __future__ import annotations

import sys
from pathlib import Path
from typing import BinaryIO

import numpy as np

from .util import DEFAULT_MAPPABILITY_READ_LENGTH, verbose_print

STDOUT_FILENAME = "-"
CHROMOSOME_FILENAME_DELIMITER = ".unique"
WIG_FIXED_STEP_DECLARATION_FORMAT = "fixedStep chrom={} start={} step=1 span=1\n"
MULTIREAD_MAPPABILITY_TYPE = np.float64
_DTYPES = {".uint8": np.uint8, ".uint16": np.uint16, ".uint32": np.uint32}


def create_multiread_mappability_from_unique_file(unique_lengths_filename: Path, kmer_length: int, data_type):
    """newmap/track.py:22-59: fraction of the k windows covering each base that are unique.
    A window starting at j counts when 0 < unique[j] <= k; base i is covered by windows
    j = i-k+1 .. i."""
    unique = np.fromfile(str(unique_lengths_filename), dtype=data_type)
    marks = ((unique <= kmer_length) & (unique != 0)).astype(np.int64)
    covered = np.cumsum(marks)
    covered[kmer_length:] -= covered[:-kmer_length].copy()          # windows that ended before i
    return covered.astype(MULTIREAD_MAPPABILITY_TYPE) / kmer_length


def write_single_read_bed(bed_file: BinaryIO, kmer_length: int, multi_read_mappability: np.ndarray, chr_name: str):
    """newmap/track.py:62-93: run-length intervals of (mappability > 0), both the 0- and the 1-runs."""
    single = multi_read_mappability > 0.0
    if single.size == 0:
        return
    starts = np.flatnonzero(np.concatenate(([True], single[1:] != single[:-1])))
    ends = np.append(starts[1:], single.size)
    values = single[starts].astype(np.int64)
    bed_file.write("".join(f"{chr_name}\t{s}\t{e}\tk{kmer_length}\t{v}\t.\n"
                           for s, e, v in zip(starts.tolist(), ends.tolist(), values.tolist())).encode())


def float_format(value: float, decimal_places: int) -> bytes:
    """newmap/track.py:114-121"""
    if value == 0.0:
        return b"0.0"
    return f"{value:.{decimal_places}f}".encode()


def write_multi_read_wig(wig_file: BinaryIO, multi_read_mappability: np.ndarray, chr_name: str,
                         decimal_places: int = 2):
    """newmap/track.py:96-111: fixedStep declaration, then one formatted value per line."""
    wig_file.write(WIG_FIXED_STEP_DECLARATION_FORMAT.format(chr_name, 1).encode())
    if multi_read_mappability.size == 0:
        return
    distinct, inverse = np.unique(multi_read_mappability, return_inverse=True)
    width = max(len(float_format(v, decimal_places)) for v in distinct.tolist()) + 1
    table = np.zeros((distinct.size, width), dtype=np.uint8)
    lengths = np.zeros(distinct.size, dtype=np.int64)
    for i, v in enumerate(distinct.tolist()):
        text = float_format(v, decimal_places) + b"\n"
        table[i, :len(text)] = np.frombuffer(text, dtype=np.uint8)
        lengths[i] = len(text)
    step = 1 << 22
    for a in range(0, inverse.size, step):                           # bounded memory
        inv = inverse[a:a + step]
        rows = table[inv]
        keep = np.arange(width)[None, :] < lengths[inv][:, None]
        wig_file.write(rows[keep].tobytes())


def safe_remove(filename):
    if filename and filename != STDOUT_FILENAME and Path(filename).exists():
        Path(filename).unlink()


def _device_for_track():
    """GPU to use; None only when the host tool was asked for explicitly (NEWMAP_AMD_TRACK=host)"""
    import os
    if os.environ.get("NEWMAP_AMD_TRACK", "") == "host":
        return None
    from .engine import default_device, device_count
    if device_count() < 1:
        raise RuntimeError("newmap track: no HIP device visible (set NEWMAP_AMD_TRACK=host to run the "
                           "host numpy tool instead)")
    return default_device()


def write_mappability_files(unique_count_filenames, kmer_length: int, single_read_bed_filename,
                            multi_read_wig_filename, verbose: bool):
    """newmap/track.py:131-237"""
    if single_read_bed_filename == STDOUT_FILENAME and multi_read_wig_filename == STDOUT_FILENAME:
        raise ValueError("Cannot output both single-read and multi-read files to standard output")
    if not single_read_bed_filename and not multi_read_wig_filename:
        raise ValueError("Must specify at least one output file")
    safe_remove(single_read_bed_filename)
    safe_remove(multi_read_wig_filename)
    for unique_path in unique_count_filenames:
        unique_path = Path(unique_path)
        base = unique_path.name
        chr_name = base[:base.find(CHROMOSOME_FILENAME_DELIMITER)]
        if unique_path.suffix not in _DTYPES:
            raise ValueError(f"Unknown extension on unique length file: \"{unique_path.suffix}\"")
        verbose_print(verbose, f"Calculating mappability regions from minimum unique k-mer lengths in "
                               f"file: {unique_path}")
        device = _device_for_track()
        if device is not None:
            import ctypes
            import os
            import tempfile
            from . import _lib
            n_pos, n_runs = ctypes.c_uint64(0), ctypes.c_uint64(0)
            with tempfile.TemporaryDirectory() as td:
                # standard output: the device writes a scratch file that is then streamed out
                bed = single_read_bed_filename
                wig = multi_read_wig_filename
                if bed == STDOUT_FILENAME:
                    bed = os.path.join(td, "stdout.bed")
                if wig == STDOUT_FILENAME:
                    wig = os.path.join(td, "stdout.wig")
                rc = _lib.lib().nm_track_file(
                    device, os.fsencode(unique_path), chr_name.encode(), np.dtype(_DTYPES[unique_path.suffix]).itemsize,
                    int(kmer_length), os.fsencode(bed) if bed else None, os.fsencode(wig) if wig else None,
                    ctypes.byref(n_pos), ctypes.byref(n_runs))
                _lib.raise_for(rc)
                for name, path in ((single_read_bed_filename, bed), (multi_read_wig_filename, wig)):
                    if name == STDOUT_FILENAME:
                        with open(path, "rb") as fh:
                            for chunk in iter(lambda: fh.read(1 << 24), b""):
                                sys.stdout.buffer.write(chunk)
                        sys.stdout.buffer.flush()
            verbose_print(verbose, "Chromosome size:")
            verbose_print(verbose, f"{chr_name}\t{n_pos.value}")
            continue
        mm = create_multiread_mappability_from_unique_file(unique_path, kmer_length, _DTYPES[unique_path.suffix])
        verbose_print(verbose, "Chromosome size:")
        verbose_print(verbose, f"{chr_name}\t{mm.shape[0]}")
        if single_read_bed_filename:
            verbose_print(verbose, "Appending single-read mappability regions to " +
                          ("standard output" if single_read_bed_filename == STDOUT_FILENAME
                           else str(single_read_bed_filename)))
            if single_read_bed_filename == STDOUT_FILENAME:
                write_single_read_bed(sys.stdout.buffer, kmer_length, mm, chr_name)
            else:
                with open(single_read_bed_filename, "ab") as fh:
                    write_single_read_bed(fh, kmer_length, mm, chr_name)
        if multi_read_wig_filename:
            verbose_print(verbose, "Appending multi-read mappability regions to " +
                          ("standard output" if multi_read_wig_filename == STDOUT_FILENAME
                           else str(multi_read_wig_filename)))
            decimal_places = int(np.ceil(np.log10(kmer_length)))     # track.py:224
            if multi_read_wig_filename == STDOUT_FILENAME:
                write_multi_read_wig(sys.stdout.buffer, mm, chr_name, decimal_places)
            else:
                with open(multi_read_wig_filename, "ab") as fh:
                    write_multi_read_wig(fh, mm, chr_name, decimal_places)


def check_unique_file_existence(filename: Path):
    if not Path(filename).exists():
        raise FileNotFoundError(f"Unique count file does not exist: {filename}")


def main(args):
    """newmap/track.py:246-284: a non-numeric first positional is one more unique file (k = 24)."""
    files = [Path(f) for f in args.unique_count_files]
    for f in files:
        check_unique_file_existence(f)
    kmer_length = args.read_length
    if not str(kmer_length).isdigit():
        extra = Path(kmer_length)
        check_unique_file_existence(extra)
        files.insert(0, extra)
        kmer_length = DEFAULT_MAPPABILITY_READ_LENGTH
    else:
        kmer_length = int(kmer_length)
    single, multi = args.single_read, args.multi_read
    if not single and not multi:
        single = STDOUT_FILENAME
    write_mappability_files(files, kmer_length, single, multi, args.verbose)

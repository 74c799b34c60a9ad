"""`newmap search` on the MI355X engine (reference: newmap/search.py).

Same entry points as the reference -- `SearchConfig`, `write_unique_counts`, `binary_search`,
`linear_search`, `get_num_kmers`, `main` -- and the same `<id>.unique.uint8|16|32` files, but the
whole per-segment loop of newmap/search.py:383-548 / :551-644 (ambiguity mask, per-position upper
bound, every count on both strands, the zero-count guard) is ONE fused device launch per segment
(`Index.min_unique_segment` / `Index.fixed_k_segment`); nothing is counted on the host.
"""
from __future__ import annotations

from contextlib import ExitStack
from dataclasses import dataclass, field
from functools import partial
from math import ceil, log2
from pathlib import Path
from typing import Callable, Sequence

from ._lazy import np        # numpy, imported at its first use: the one-shot CLI's native path never needs it (0.15 s)

from .engine import Index, cached_index, cached_indexes
from .fasta import SequenceSegment, sequence_segments
from .util import INDEX_EXTENSION, optional_gzip_open, verbose_print

KMER_RANGE_SEPARATOR = ":"
SEQUENCE_ID_SEPARATOR = ","
INDEX_FILE_SEPARATOR = ","
FASTA_FILE_SEPARATOR = ","
UNIQUE_COUNT_FILENAME_FORMAT = "{}.unique.{}"


def _no_log(*_):
    pass


@dataclass(frozen=True)
class SearchConfig:
    """Field-for-field the reference's SearchConfig (newmap/search.py:32-63) plus `device`."""
    fasta_filepaths: list
    fmindex_filepaths: list
    kmer_lengths: list
    is_binary_search: bool
    use_reverse_complement: bool = True
    output_directory: Path = field(default_factory=Path.cwd)
    include_sequence_ids: list = field(default_factory=list)
    exclude_sequence_ids: list = field(default_factory=list)
    num_threads: int = 1                 # accepted for interface parity; parallelism is the GPU's
    kmer_batch_size: int = 1000000
    initial_search_length: int = 0
    verbose: bool = False
    device: int | None = None            # engine extension: which MI355X (default LOCAL_RANK or 0)
    log: Callable = field(init=False, default=_no_log)

    @classmethod
    def from_args(cls, args):
        """newmap/search.py:65-184: flags -> config (same parsing rules and error messages)."""
        out_dir = Path(args.output_directory) if args.output_directory else Path(".")
        out_dir.mkdir(parents=True, exist_ok=True)
        fastas = [Path(f) for f in args.fasta_file.split(FASTA_FILE_SEPARATOR)]
        if not args.index_file:
            indexes = [Path(fastas[0].stem + "." + INDEX_EXTENSION)]
        else:
            indexes = [Path(f) for f in args.index_file.split(INDEX_FILE_SEPARATOR)]
        for ix in indexes:
            if not ix.is_file():
                raise FileNotFoundError(f"Index file not found: {ix}")
        spec = args.search_range
        if KMER_RANGE_SEPARATOR in spec:
            try:
                lo, hi = map(int, spec.split(KMER_RANGE_SEPARATOR))
            except ValueError:
                raise ValueError("Could not parse k-mer search range format") from None
            if lo > hi:
                raise ValueError("K-mer range start length is larger than the end length")
            lengths, binary = list(range(lo, hi + 1)), True
        else:
            lengths, binary = [int(x) for x in spec.split(",")], False
        if args.initial_search_length and not binary:
            raise ValueError("Initial search length only valid when a range of k-mer lengths is given")
        if args.include_sequences and args.exclude_sequences:
            raise ValueError("Cannot specify both include and exclude sequences")
        include = [s.encode() for s in args.include_sequences.split(SEQUENCE_ID_SEPARATOR)] \
            if args.include_sequences else []
        exclude = [s.encode() for s in args.exclude_sequences.split(SEQUENCE_ID_SEPARATOR)] \
            if args.exclude_sequences else []
        return cls(fasta_filepaths=fastas, fmindex_filepaths=indexes, kmer_lengths=lengths,
                   is_binary_search=binary, use_reverse_complement=not args.norc,
                   output_directory=out_dir, include_sequence_ids=include, exclude_sequence_ids=exclude,
                   verbose=args.verbose, num_threads=args.num_threads, kmer_batch_size=args.kmer_batch_size,
                   initial_search_length=args.initial_search_length,
                   device=getattr(args, "device", None))

    def __post_init__(self):
        object.__setattr__(self, "log", partial(verbose_print, True) if self.verbose else _no_log)


def output_type(max_kmer_length: int):
    """newmap/search.py:204-212"""
    if max_kmer_length <= np.iinfo(np.uint8).max:
        return np.uint8, "uint8"
    if max_kmer_length <= np.iinfo(np.uint16).max:
        return np.uint16, "uint16"
    return np.uint32, "uint32"


def get_num_kmers(sequence_segment: SequenceSegment, max_kmer_length: int) -> int:
    """newmap/search.py:727-741"""
    if sequence_segment.epilogue:
        return len(sequence_segment.data)
    return len(sequence_segment.data) - (max_kmer_length - 1)


def _index(config: SearchConfig) -> Index:
    return cached_index(config.fmindex_filepaths[0], config.device)


def binary_search(config: SearchConfig, sequence_segments: Sequence[SequenceSegment], min_kmer_length: int,
                  max_kmer_length: int, data_type):
    """newmap/search.py:383-548 -> (unique_lengths, ambiguous_positions_skipped), one launch."""
    if len(sequence_segments) > 1 or len(config.fmindex_filepaths) > 1:
        from .engine import search_segment_multi
        indexes = cached_indexes(config.fmindex_filepaths, config.device)
        return search_segment_multi(indexes, [s.data for s in sequence_segments],
                                    get_num_kmers(sequence_segments[0], max_kmer_length),
                                    [min_kmer_length, max_kmer_length], True, config.use_reverse_complement, data_type)
    seg = sequence_segments[0]
    ix = _index(config)
    num_kmers = get_num_kmers(seg, max_kmer_length)
    unique, n_amb = ix.min_unique_segment(seg.data, num_kmers, min_kmer_length, max_kmer_length,
                                          config.use_reverse_complement, data_type,
                                          config.initial_search_length)
    config.log(f"Skipping {n_amb} ambiguous positions")
    return unique, n_amb


def linear_search(config: SearchConfig, sequence_segments: Sequence[SequenceSegment], num_kmers: int, data_type):
    """newmap/search.py:551-644 -> (unique_lengths, ambiguous_positions_skipped), one launch."""
    if len(sequence_segments) > 1 or len(config.fmindex_filepaths) > 1:
        from .engine import search_segment_multi
        indexes = cached_indexes(config.fmindex_filepaths, config.device)
        return search_segment_multi(indexes, [s.data for s in sequence_segments], num_kmers, config.kmer_lengths,
                                    False, config.use_reverse_complement, data_type)
    seg = sequence_segments[0]
    ix = _index(config)
    unique, n_amb = ix.fixed_k_segment(seg.data, num_kmers, config.kmer_lengths, config.use_reverse_complement,
                                       data_type)
    config.log(f"Skipping {n_amb} ambiguous positions")
    return unique, n_amb


class _Summary:
    """running per-run statistics printed by newmap/search.py:885-902"""

    def __init__(self, kmin, kmax):
        self.ambiguous = self.unique = self.none = 0
        self.max_len, self.min_len = kmin, kmax

    def add(self, arr: np.ndarray, num_kmers: int, n_amb: int):
        n_unique = int(np.count_nonzero(arr))
        self.ambiguous += n_amb
        self.unique += n_unique
        self.none += num_kmers - n_unique - n_amb
        if n_unique:
            self.max_len = max(self.max_len, int(arr.max()))
            self.min_len = min(self.min_len, int(arr[arr != 0].min()))
        return n_unique

    def report(self, config, sequence_id: bytes):
        if self.unique:
            config.log(f"Finished writing unique lengths for sequence ID: {sequence_id.decode()}")
            config.log(f"{self.unique} unique lengths found")
            config.log(f"{self.ambiguous} positions skipped due to ambiguity")
            config.log(f"{self.none} positions with no unique length found")
            config.log(f"{self.max_len}-mer maximum unique length found")
            config.log(f"{self.min_len}-mer minimum unique length found")


def _wanted(config: SearchConfig, sequence_id: bytes) -> bool:
    """--include-sequences / --exclude-sequences.  Intended semantics; the reference's test at
    newmap/search.py:270-278 makes --exclude-sequences alone skip EVERY record (then raise
    :376-380) -- a conscious divergence, see DESIGN.md."""
    if config.include_sequence_ids:
        return sequence_id in config.include_sequence_ids
    if config.exclude_sequence_ids:
        return sequence_id not in config.exclude_sequence_ids
    return True


def write_unique_counts(config: SearchConfig):
    """newmap/search.py:197-380: one `<id>.unique.<dtype>` file per FASTA record.

    Runs the native driver (csrc/nm_driver.hip: streaming FASTA reader, pinned double buffers, copies and
    kernels overlapped with parsing and file appends); NEWMAP_AMD_PYTHON_DRIVER=1 selects the
    segment-by-segment Python loop below, which produces the same files."""
    import os
    if len(config.fasta_filepaths) > 1 or len(config.fmindex_filepaths) > 1:
        return _write_unique_counts_multi(config)
    if os.environ.get("NEWMAP_AMD_PYTHON_DRIVER", "") != "1":
        return _write_unique_counts_native(config)
    return _write_unique_counts_python(config)


def _check_range(config: SearchConfig, min_kmer_length: int, max_kmer_length: int):
    if config.is_binary_search:
        # NB: the reference evaluates log2(kmax - kmin) here and so rejects a:a ranges with
        # "math domain error" (newmap/search.py:215-217); same exception, clearer message.
        if max_kmer_length == min_kmer_length:
            raise ValueError("math domain error: a k-mer range needs two different lengths")
        config.log("Max {} iterations over range {}-{}".format(
            ceil(log2(max_kmer_length - min_kmer_length) + 1), min_kmer_length, max_kmer_length))


def _nothing_processed(config: SearchConfig):
    if config.include_sequence_ids:                                   # newmap/search.py:368-380
        raise ValueError(f"None of the included sequences were found: {config.include_sequence_ids}")
    if config.exclude_sequence_ids:
        raise ValueError("The excluded sequences were too strict and nothing was processed: "
                         f"{config.exclude_sequence_ids}")


def _write_unique_counts_native(config: SearchConfig):
    max_kmer_length, min_kmer_length = max(config.kmer_lengths), min(config.kmer_lengths)
    _check_range(config, min_kmer_length, max_kmer_length)
    index = _index(config)
    index.set_initial_search_length(config.initial_search_length)      # (the exact guard replays the reference's schedule)
    running = _Summary(min_kmer_length, max_kmer_length)

    def on_record(rec_id: bytes, s: dict):
        config.log(f"Writing unique lengths for sequence ID: {rec_id.decode()}")
        running.unique += s["unique"]
        running.ambiguous += s["ambiguous"]
        running.none += s["no_unique"]
        if s["unique"]:
            running.max_len = max(running.max_len, s["max_len"])
            running.min_len = min(running.min_len, s["min_len"])
        running.report(config, rec_id)

    total = index.search_fasta(config.fasta_filepaths[0], config.output_directory, config.kmer_lengths,
                               config.is_binary_search, config.use_reverse_complement, config.kmer_batch_size,
                               config.include_sequence_ids, config.exclude_sequence_ids,
                               on_record if config.verbose else None)
    if total["records"] == 0:
        _nothing_processed(config)


def _write_unique_counts_multi(config: SearchConfig):
    """Several FASTA files in lock-step and/or several index files (newmap/search.py:251-265, 461,
    656-697): ids, ambiguity mask and upper bounds come from the first FASTA; the count of a position
    is summed over every (index, sequence) pair (`nm_search_segment_multi`)."""
    from .engine import search_segment_multi
    max_kmer_length, min_kmer_length = max(config.kmer_lengths), min(config.kmer_lengths)
    data_type, suffix = output_type(max_kmer_length)
    _check_range(config, min_kmer_length, max_kmer_length)
    indexes = cached_indexes(config.fmindex_filepaths, config.device)
    lookahead = max_kmer_length - 1
    requested = config.kmer_batch_size + lookahead
    processed_any = False
    summary = _Summary(min_kmer_length, max_kmer_length)
    current_id, current_path = None, None
    with ExitStack() as stack:
        files = [stack.enter_context(optional_gzip_open(p, "rb")) for p in config.fasta_filepaths]
        streams = [sequence_segments(f, requested, lookahead) for f in files]
        for segs in zip(*streams):                                    # :260
            seg = segs[0]                                             # :265
            if seg.id != current_id:
                if not _wanted(config, seg.id):
                    continue
                if current_id is not None:
                    summary.report(config, current_id)
                processed_any = True
                current_id = seg.id
                current_path = Path(config.output_directory) / UNIQUE_COUNT_FILENAME_FORMAT.format(
                    seg.id.decode(), suffix)
                open(current_path, "wb").close()
                config.log(f"Writing unique lengths for sequence ID: {seg.id.decode()}")
            num_kmers = get_num_kmers(seg, max_kmer_length)
            arr, n_amb = search_segment_multi(indexes, [s.data for s in segs], num_kmers, config.kmer_lengths,
                                              config.is_binary_search, config.use_reverse_complement, data_type)
            summary.add(arr, num_kmers, n_amb)
            with open(current_path, "ab") as fh:
                arr.tofile(fh)
        if current_id is not None:
            summary.report(config, current_id)
    if not processed_any:
        _nothing_processed(config)


def _write_unique_counts_python(config: SearchConfig):
    """the same driver, one segment at a time from Python (reference-shaped loop)"""
    max_kmer_length = max(config.kmer_lengths)
    min_kmer_length = min(config.kmer_lengths)
    data_type, suffix = output_type(max_kmer_length)
    _check_range(config, min_kmer_length, max_kmer_length)
    index = _index(config)

    lookahead = max_kmer_length - 1                                   # :229
    requested = config.kmer_batch_size + lookahead                    # :235
    processed_any = False
    summary = _Summary(min_kmer_length, max_kmer_length)
    current_id, current_path = None, None
    # the record being searched: its fingerprint joined from the segments' (csrc/nm_hash.h).  A record that is not one of the
    # indexed ones goes through the exact guard (newmap/search.py:699-722) in a second pass over the FASTA, as in the native
    # driver: nothing of a record is kept beyond its current segment (--kmer-batch-size bounds the memory, as in the reference)
    rec_fp, rec_pos, rec_joinable, rec_no = 0, 0, True, 0
    unverified: set = set()

    def end_of_record():
        nonlocal rec_fp, rec_pos, rec_joinable
        if rec_pos and not (rec_joinable and index.has_record(rec_pos, rec_fp)):
            unverified.add(rec_no)
        rec_fp, rec_pos, rec_joinable = 0, 0, True

    with ExitStack() as stack:
        fasta = stack.enter_context(optional_gzip_open(config.fasta_filepaths[0], "rb"))
        index.set_segment_guard(False)                                # (whole records are checked below, not every segment)
        stack.callback(index.set_segment_guard, True)
        for seg in sequence_segments(fasta, requested, lookahead):
            if seg.id != current_id:
                if not _wanted(config, seg.id):
                    rec_no += int(seg.epilogue)                       # (records are numbered as they come, wanted or not)
                    continue
                if current_id is not None:
                    summary.report(config, current_id)
                processed_any = True
                current_id = seg.id
                current_path = Path(config.output_directory) / UNIQUE_COUNT_FILENAME_FORMAT.format(
                    seg.id.decode(), suffix)
                open(current_path, "wb").close()                      # :304-305 truncate on a new id
                config.log(f"Writing unique lengths for sequence ID: {seg.id.decode()}")
            num_kmers = get_num_kmers(seg, max_kmer_length)
            config.log(f"Processing {num_kmers} k-mers")
            if config.is_binary_search:
                arr, n_amb = index.min_unique_segment(seg.data, num_kmers, min_kmer_length, max_kmer_length,
                                                      config.use_reverse_complement, data_type,
                                                      config.initial_search_length)
            else:
                arr, n_amb = index.fixed_k_segment(seg.data, num_kmers, config.kmer_lengths,
                                                   config.use_reverse_complement, data_type)
            if summary.add(arr, num_kmers, n_amb) == 0:
                config.log("No unique lengths found for this sequence segment")
            with open(current_path, "ab") as fh:                      # :356-357
                arr.tofile(fh)
            if rec_pos % 64:
                rec_joinable = False
            else:
                from .engine import fingerprint_join
                rec_fp = (rec_fp + fingerprint_join(0, rec_pos, index.last_fingerprint())) & 0xFFFFFFFFFFFFFFFF
            rec_pos += num_kmers
            if seg.epilogue:
                end_of_record()
                rec_no += 1
        if current_id is not None:
            summary.report(config, current_id)
        if unverified:                                                # the exact guard, record by record, segment by segment
            fasta.seek(0)
            n, cur_id = 0, None
            for seg in sequence_segments(fasta, requested, lookahead):
                # (the first pass skips the later segments of an unwanted id the same way: `seg.id != current_id` stays true for them)
                wanted = seg.id == cur_id or _wanted(config, seg.id)
                if wanted:
                    cur_id = seg.id
                    if n in unverified:
                        index.guard_segment(seg.data, get_num_kmers(seg, max_kmer_length), config.kmer_lengths, config.is_binary_search,
                                            config.use_reverse_complement, config.initial_search_length)
                n += int(seg.epilogue)
    if not processed_any:                                             # :368-380
        _nothing_processed(config)


def main(args):
    import os
    config = SearchConfig.from_args(args)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # launched one rank per GPU (python -m torch.distributed.run ... -m newmap_amd.main search ...)
        from .parallel import write_unique_counts_distributed
        write_unique_counts_distributed(config)
    else:
        write_unique_counts(config)

"""FASTA records and segment geometry of the `search` path (reference: newmap/fasta.py).

Same interface (`SequenceSegment`, `sequence_segments`) and the same bytes in every segment as
the reference's line-streaming generator (newmap/fasta.py:20-190; pinned by
tests/golden/golden_host.json), but built block-wise: the file is read in large blocks, header
lines are located with one regular-expression scan, and sequence blocks lose their line ends in
one `bytes.translate`, so a 3 Gbp genome does not pay a Python iteration per 60-base line.
"""
from __future__ import annotations

import re
from typing import BinaryIO, Iterator

FASTA_FILE_IGNORE_DELIMITERS = (b">", b";")            # newmap/fasta.py:4
_HEADER_RE = re.compile(rb"^[>;][^\n]*", re.MULTILINE)
_ODD_SPACE_RE = re.compile(rb"[ \t\v\f]|\r(?!\n)")
_BLOCK_BYTES = 64 << 20


class SequenceSegment:
    """newmap/fasta.py:7-17, plus `offset` (where data[0] sits in its record)."""

    __slots__ = ("id", "data", "epilogue", "offset")

    def __init__(self, sequence_id: bytes, data: bytes = b"", epilogue: bool = False, offset: int = 0):
        self.id = sequence_id
        self.data = data
        self.epilogue = epilogue
        self.offset = offset

    def is_empty(self):
        return len(self.data) == 0


def _clean(block: bytes) -> bytes:
    """Sequence lines of `block`, each right-stripped (newmap/fasta.py:47), concatenated."""
    if not block:
        return b""
    if _ODD_SPACE_RE.search(block) is None:
        return block.translate(None, b"\r\n")            # only line ends to remove
    return b"".join(line.rstrip() for line in block.split(b"\n"))


def fasta_records(fasta_file: BinaryIO) -> Iterator[tuple[bytes, bytes]]:
    """Yield (record id, sequence bytes) for every record that has data.

    '>' or ';' at a line start opens a record (newmap/fasta.py:59); its id is the first
    whitespace-delimited token without its first byte (:75); data in front of any header
    belongs to id b''; a header without data yields nothing (:173-188)."""
    cur_id, parts = b"", []
    carry = b""

    def flush():
        data = b"".join(parts)
        parts.clear()
        if data:
            yield cur_id, data

    while True:
        block = fasta_file.read(_BLOCK_BYTES)
        if not block:
            break
        block = carry + block
        cut = block.rfind(b"\n")
        if cut < 0:
            carry = block
            continue
        carry, block = block[cut + 1:], block[:cut + 1]
        pos = 0
        for m in _HEADER_RE.finditer(block):
            parts.append(_clean(block[pos:m.start()]))
            yield from flush()
            cur_id = m.group().rstrip().split()[0][1:]
            pos = m.end()
        parts.append(_clean(block[pos:]))
    if carry:                                            # last line without a newline
        if carry.startswith(FASTA_FILE_IGNORE_DELIMITERS):
            yield from flush()
            cur_id = carry.rstrip().split()[0][1:]
        else:
            parts.append(carry.rstrip())
    yield from flush()


def record_segments(sequence_id: bytes, data: bytes, sequence_length: int,
                    sequence_overlap_length: int = 0) -> Iterator[SequenceSegment]:
    """Segments of one record: [j*B, j*B + length) with B = length - overlap while they fit, then
    the remainder; the last one carries `epilogue` (newmap/fasta.py:109-190)."""
    n = len(data)
    if n == 0:
        return
    step = sequence_length - sequence_overlap_length
    if step <= 0:
        raise ValueError("sequence length must exceed the overlap length")
    start = 0
    while start + sequence_length < n:
        yield SequenceSegment(sequence_id, data[start:start + sequence_length], False, start)
        start += step
    # what is left is either an exact fit or a shorter tail: both are the epilogue
    yield SequenceSegment(sequence_id, data[start:], True, start)


def sequence_segments(fasta_file: BinaryIO, sequence_length: int,
                      sequence_overlap_length: int = 0) -> Iterator[SequenceSegment]:
    """newmap/fasta.py:20: iterate over a FASTA file, yielding SequenceSegments."""
    for sequence_id, data in fasta_records(fasta_file):
        yield from record_segments(sequence_id, data, sequence_length, sequence_overlap_length)

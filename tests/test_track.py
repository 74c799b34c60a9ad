"""`track`, the HOST numpy tool (selected explicitly with NEWMAP_AMD_TRACK=host): byte-identical BED / WIG
to the reference's newmap/track.py on fixtures produced by the reference itself
(tests/golden/make_golden_track.py).  The device implementation is checked against the same fixtures
in tests/test_gpu_parity.py."""
import argparse
import json
from pathlib import Path

import numpy as np
import pytest

GOLDEN = Path(__file__).resolve().parent / "golden" / "golden_track.json"


@pytest.fixture(autouse=True)
def _host_tool(monkeypatch):
    monkeypatch.setenv("NEWMAP_AMD_TRACK", "host")


@pytest.fixture(scope="module")
def golden():
    return json.loads(GOLDEN.read_text())


def _write_arrays(tmp_path, golden):
    files = []
    for name, a in golden["arrays"].items():
        p = tmp_path / f"{name}.unique.{a['dtype']}"
        np.array(a["values"], dtype=a["dtype"]).tofile(p)
        files.append(p)
    return files


def test_track_files_match_reference(tmp_path, golden):
    from newmap_amd.track import write_mappability_files
    files = _write_arrays(tmp_path, golden)
    for c in golden["cases"]:
        bed, wig = tmp_path / "o.bed", tmp_path / "o.wig"
        write_mappability_files(files, c["k"], str(bed), str(wig), False)
        assert bed.read_text() == c["bed"], c["k"]
        assert wig.read_text() == c["wig"], c["k"]


def test_track_main_defaults_and_errors(tmp_path, golden, capsysbinary):
    from newmap_amd import track
    files = _write_arrays(tmp_path, golden)
    # a non-numeric first positional is one more unique file and k defaults to 24
    args = argparse.Namespace(read_length=str(files[0]), unique_count_files=[str(f) for f in files[1:]],
                              single_read=None, multi_read=None, verbose=False)
    track.main(args)
    out = capsysbinary.readouterr().out.decode()
    want = [c for c in golden["cases"] if c["k"] == 24][0]["bed"]
    assert out == want
    with pytest.raises(ValueError, match="both single-read and multi-read"):
        track.write_mappability_files(files, 10, "-", "-", False)
    with pytest.raises(ValueError, match="at least one output"):
        track.write_mappability_files(files, 10, None, None, False)
    with pytest.raises(FileNotFoundError):
        track.main(argparse.Namespace(read_length="10", unique_count_files=[str(tmp_path / "nope.unique.uint8")],
                                      single_read=None, multi_read=None, verbose=False))
    bad = tmp_path / "x.unique.int8"
    bad.write_bytes(b"\0" * 4)
    with pytest.raises(ValueError, match="Unknown extension"):
        track.write_mappability_files([bad], 10, str(tmp_path / "b.bed"), None, False)


def test_track_without_gpu_fails_loudly(tmp_path, golden, monkeypatch):
    from newmap_amd import track
    from newmap_amd.engine import device_count
    if device_count() > 0:
        pytest.skip("a GPU is visible")
    monkeypatch.delenv("NEWMAP_AMD_TRACK")
    files = _write_arrays(tmp_path, golden)
    with pytest.raises(RuntimeError, match="no HIP device"):
        track.write_mappability_files(files, 10, str(tmp_path / "x.bed"), None, False)

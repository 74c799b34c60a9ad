"""AddressSanitizer + UBSan runs of the native HOST code (GPU sanitizers are not available on the
pool): the oracle (oracle/kmer_oracle.c) and the product's index builder (csrc/nm_build.cpp with both
suffix sorters).  Each scenario runs in a child Python with libasan preloaded."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "hostsim" / "_build"


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if out and Path(out).exists() else None


def _run(code: str, extra_env=None):
    asan = _libasan()
    if asan is None:
        pytest.skip("libasan not found")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="3")
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_oracle_under_asan_ubsan():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "-s", "_build/liboracle_asan.so"], check=True)
    code = f"""
import sys, ctypes, numpy as np
sys.path.insert(0, {str(ROOT)!r})
from oracle import ref_driver as rd
rd._LIB_PATH = rd._HERE / "_build" / "liboracle_asan.so"
rng = np.random.default_rng(3)
a = bytes(np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, 5000, p=[.24,.24,.24,.24,.04])])
b = b"ACGT" * 300 + b"A" * 500 + a[100:900]
ix = rd.OracleIndex([a, b, b"A"])
ix.enable_fm(5)
for fm in (False, True):
    got = rd.unique_counts([b">a\\n", a + b"\\n", b">b\\n", b + b"\\n"], ix, list(range(6, 41)), True, 700, True, 0, fm)
    assert got[b"a"].size == len(a)
    rd.ref_binary_search_segment_c(ix, a, len(a), 6, 40, fm=fm)
lin = rd.unique_counts([b">a\\n", a + b"\\n"], ix, [30, 12], False, 10**6)
print("ok")
"""
    assert "ok" in _run(code)


def test_index_builder_under_asan_ubsan(tmp_path):
    BUILD.mkdir(exist_ok=True)
    so = BUILD / "libnmbuild_asan.so"
    src = ROOT / "newmap_amd" / "csrc" / "nm_build.cpp"
    deps = [src, src.with_name("nm_sais.hpp"), src.with_name("nm_pdsa.hpp")]
    if not so.exists() or so.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-fsanitize=address,undefined",
                        "-fno-omit-frame-pointer", "-fPIC", "-shared", "-o", str(so), str(src), "-lz"], check=True)
    fa = tmp_path / "s.fa"
    import numpy as np
    rng = np.random.default_rng(9)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    body = bytes(alpha[rng.integers(0, 4, 90_000)])
    fa.write_bytes(b"hdrless\n>a x\n" + body[:40_000] + b"\nNNNN" + b"AC" * 5000 + b"\n>b\n\n>c\n" + b"T" * 20_000 +
                   body[40_000:] + b"\n;d\r\nacgtnACGT\r\n>e\nN\n")
    code = f"""
import ctypes
L = ctypes.CDLL({str(so)!r})
L.nm_index_build.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint8, ctypes.c_uint8]
L.nm_last_error.restype = ctypes.c_char_p
rc = L.nm_index_build({str(fa)!r}.encode(), {str(tmp_path / 'o.awfmi')!r}.encode(), 8, 12)
assert rc == 0, L.nm_last_error()
assert L.nm_index_build(b"/nonexistent.fa", b"/tmp/x", 8, 12) == 1
print("ok")
"""
    out = {}
    for algo in ("sais", "pd"):
        assert "ok" in _run(code, {"NEWMAP_AMD_SA": algo})
        out[algo] = (tmp_path / "o.awfmi").read_bytes()
    assert out["sais"] == out["pd"]

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


def _libtsan():
    out = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    return out if out and Path(out).exists() else None


def test_native_driver_host_logic_under_tsan(tmp_path):
    """The worker pool of newmap_amd/csrc/nm_driver.hip -- units, pinned slots, hand-over to the device, pwrite at offsets,
    per-file statistics, record fingerprints, guard pass -- compiled for the host against a stand-in HIP runtime
    (tests/hostsim/fake_hip) and stubbed engine calls (tests/hostsim/driver_sim.cpp), run under ThreadSanitizer with 8
    workers: no data race, the engine is never entered by two workers at once, and the files equal a Python model of
    the stub's function -- for batches fused or not, one rank and three, a skipped record between two records of one
    id (appends, as newmap/search.py:268-305), a record that is not indexed (guard pass) and one the guard rejects."""
    tsan = _libtsan()
    if tsan is None:
        pytest.skip("libtsan not found")
    BUILD.mkdir(exist_ok=True)
    so = BUILD / "libdriver_tsan.so"
    src = ROOT / "newmap_amd" / "csrc" / "nm_driver.hip"
    sim = ROOT / "tests" / "hostsim" / "driver_sim.cpp"
    deps = [src, sim, src.with_name("nm_fasta_scan.hpp"), src.with_name("nm_hash.h"), ROOT / "tests" / "hostsim" / "fake_hip" / "hip" / "hip_runtime.h"]
    if not so.exists() or so.stat().st_mtime < max(d.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-fPIC", "-shared", "-pthread", "-DNM_DRIVER_HOST_SUMMARY",
                        f"-I{ROOT / 'tests' / 'hostsim' / 'fake_hip'}", "-x", "c++", str(src), str(sim), "-o", str(so), "-lz"], check=True)
    code = f"""
import ctypes, os, sys
import numpy as np
L = ctypes.CDLL({str(so)!r})
vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
L.ds_index_new.restype = vp
L.ds_index_add_record.argtypes = [vp, ctypes.c_char_p, u64]
L.nm_index_info.restype = u64
L.nm_index_info.argtypes = [vp, i32]
L.nm_last_error.restype = ctypes.c_char_p
L.nm_search_fasta_shard.restype = i32
L.nm_search_fasta_shard.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, vp, u32, i32, i32, u64, ctypes.POINTER(ctypes.c_char_p), u32,
                                    ctypes.POINTER(ctypes.c_char_p), u32, vp, vp, vp, i32, i32]
rng = np.random.default_rng(4)
alpha = np.frombuffer(b"ACGTN", np.uint8)
def dna(n): return bytes(alpha[rng.choice(5, n, p=[.24, .24, .24, .24, .04])])
recs = [(b"", dna(700)), (b"a", dna(200_003)), (b"skip", dna(5000)), (b"a", dna(64 * 300)), (b"b", dna(1)), (b"long", dna(150_000)), (b"c", dna(90_001))]
def fasta(records, width=61):
    out = b""
    for rid, d in records:
        if rid: out += b">" + rid + b" x y\\n"
        out += (d + b"\\n") if rid == b"long" else b"".join(d[i:i + width] + b"\\r\\n" for i in range(0, len(d), width))
        if rid == b"b": out += b">nodata\\n"
    return out
tmp = {str(tmp_path)!r}
fa = os.path.join(tmp, "g.fa")
open(fa, "wb").write(fasta(recs))
ix = L.ds_index_new()
for rid, d in recs: L.ds_index_add_record(ix, d, len(d))
def model(d, kmax):
    a = np.frombuffer(d, np.uint8).astype(np.int64)
    v = np.zeros(len(d), np.int64)
    n = len(d) - (kmax - 1)
    if n > 0: v[:n] = 1 + (a[:n] * 7 + a[kmax - 1:kmax - 1 + n] * 13 + kmax) % 200
    v[~np.isin(a, np.frombuffer(b"ACGTacgt", np.uint8))] = 0
    return v.astype(np.uint8)
def run(name, ks, is_range, batch, world=1, exclude=(b"skip",), fasta_path=fa):
    out = os.path.join(tmp, name); os.makedirs(out, exist_ok=True)
    k = np.asarray(ks, np.uint32)
    exc = (ctypes.c_char_p * max(len(exclude), 1))(*exclude)
    for rank in range(world):
        rc = L.nm_search_fasta_shard(ix, fasta_path.encode(), out.encode(), k.ctypes.data, k.size, int(is_range), 1, batch, None, 0, exc, len(exclude), None, None, None, rank, world)
        if rc: return rc, L.nm_last_error().decode()
    return 0, out
os.environ["NEWMAP_AMD_DRIVER_SLOTS"] = "8"
for fuse in ("0", "1"):
    os.environ["NEWMAP_AMD_DRIVER_FUSE"] = fuse
    for ks, is_range, batch, world in (([20, 200], True, 64 * 100, 1), ([36], False, 5000, 1), ([20, 60], True, 64 * 37, 3), ([20, 200], True, 10_000_000, 1), ([36], False, 50, 1)):
        if world > 1: os.environ["NEWMAP_AMD_SHARD_CHUNK"] = "30000"
        rc, out = run(f"o_{{fuse}}_{{batch}}_{{world}}", ks, is_range, batch, world)
        os.environ.pop("NEWMAP_AMD_SHARD_CHUNK", None)
        assert rc == 0, out
        kmax = max(ks)
        want = {{b"": model(recs[0][1], kmax), b"a": np.concatenate([model(recs[1][1], kmax), model(recs[3][1], kmax)]), b"b": model(recs[4][1], kmax),
                b"long": model(recs[5][1], kmax), b"c": model(recs[6][1], kmax)}}
        assert sorted(os.listdir(out)) == sorted((k_.decode() + ".unique.uint8") for k_ in want), os.listdir(out)
        for rid, w in want.items():
            got = np.fromfile(os.path.join(out, rid.decode() + ".unique.uint8"), np.uint8)
            assert np.array_equal(got, w), (fuse, ks, batch, world, rid)
assert L.nm_index_info(ix, 23) > 0       # (a batch below 64 cannot be rounded to a multiple of 64 and its segments do not join: those records went through the guard; 5000 works like 4992)
before = L.nm_index_info(ix, 23)
os.environ["NEWMAP_AMD_DRIVER_FUSE"] = "1"
rc, out = run("same", [20, 200], True, 64 * 1000)
assert rc == 0 and L.nm_index_info(ix, 23) == before      # every record is an indexed one: no guard
# a record that is not indexed: guarded; with an 'X' the stub's guard rejects it
other = list(recs); other[5] = (b"long", recs[5][1][:70_000] + b"A" + recs[5][1][70_001:] if recs[5][1][70_000:70_001] != b"A" else recs[5][1][:70_000] + b"C" + recs[5][1][70_001:])
fb = os.path.join(tmp, "h.fa"); open(fb, "wb").write(fasta(other))
rc, out = run("other", [20, 200], True, 64 * 1000, fasta_path=fb)
assert rc == 0 and L.nm_index_info(ix, 23) > before
bad = list(recs); bad[6] = (b"c", recs[6][1][:500] + b"X" + recs[6][1][501:])
fc = os.path.join(tmp, "x.fa"); open(fc, "wb").write(fasta(bad))
rc, msg = run("bad", [20, 200], True, 64 * 1000, fasta_path=fc)
assert rc == 8 and "not found in the index" in msg and "'c'" in msg, (rc, msg)
print("ok")
"""
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="halt_on_error=1:exitcode=66:report_signal_unsafe=0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])

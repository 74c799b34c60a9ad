"""ctypes binding of libnewmap_amd.so (include/newmap_amd.h).  Fails loudly when the library
is missing -- the package has no other compute path."""
from __future__ import annotations

import ctypes
import os
import sys
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# streams of one process share GPU_MAX_HW_QUEUES hardware queues (HIP's default: 4); the engine's lanes, side streams and the
# native driver's five streams overlap only when they land on different queues (bench.py).  Takes effect
# when set before the process's first HIP call; a caller's own setting is kept.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
LIB_PATH = Path(os.environ["NEWMAP_AMD_LIB"]) if os.environ.get("NEWMAP_AMD_LIB") else _HERE / "libnewmap_amd.so"    # (NEWMAP_AMD_LIB: tools/ load the measurement build)

NM_OK = 0
NM_E_FILE_OPEN, NM_E_ALLOC, NM_E_FILE_EXISTS, NM_E_FILE_WRITE, NM_E_FILE_FORMAT = 1, 2, 3, 4, 5
NM_E_ARGUMENT, NM_E_DEVICE, NM_E_KMER_NOT_FOUND, NM_E_TOO_LARGE = 6, 7, 8, 9
NM_STATUS_WORDS = 16
NM_STATUS_HASH = 8
NM_OPT_COUNT_STEPS = 1
NM_OPT_TIMING = 3
NM_OPT_KERNEL = 4
NM_OPT_FORCE_BIG = 6
NM_OPT_SEED_POLICY = 7
NM_OPT_LF_BLOCKS = 9
NM_OPT_REPEAT_PROBES = 10
NM_OPT_LIST_VIA_RANGE = 11
NM_OPT_SITE_D = 12
NM_OPT_SITE_TABLE = 13
NM_OPT_INITIAL_LENGTH = 14
NM_OPT_SEGMENT_GUARD = 15
NM_OPT_LF2 = 16
NM_OPT_SWEEP = 17
NM_OPT_LCP = 18

EXPORTS = [
    "nm_last_error", "nm_version", "nm_index_build", "nm_index_open", "nm_index_open_budget", "nm_dev_free_bytes", "nm_index_close",
    "nm_index_info", "nm_count_kmers", "nm_count_from_sequence", "nm_min_unique_segment",
    "nm_fixed_k_segment", "nm_upper_bound_segment", "nm_min_unique_segment_dev",
    "nm_fixed_k_segment_dev", "nm_set_option", "nm_dev_alloc", "nm_dev_free", "nm_dev_upload",
    "nm_dev_download", "nm_dev_sync", "nm_device_count", "nm_timing_read", "nm_timing_read_kind", "nm_search_fasta", "nm_search_fasta_shard", "nm_track_file", "nm_search_segment_multi", "nm_index_build_device",
    "nm_index_has_record", "nm_index_records", "nm_fingerprint_join", "nm_fingerprint_sequence", "nm_guard_segment_dev", "nm_guard_segment",
    "nm_search_fasta_shard_ex", "nm_guard_fasta", "nm_stream_release",
]

_lib = None


class SearchSummary(ctypes.Structure):
    _fields_ = [("records", ctypes.c_uint64), ("positions", ctypes.c_uint64), ("ambiguous", ctypes.c_uint64),
                ("unique", ctypes.c_uint64), ("no_unique", ctypes.c_uint64), ("max_len", ctypes.c_uint32),
                ("min_len", ctypes.c_uint32)]


RECORD_CALLBACK = ctypes.CFUNCTYPE(None, ctypes.c_char_p, ctypes.POINTER(SearchSummary), ctypes.c_void_p)


class EngineMissingError(ImportError):
    pass


def build(verbose: bool = False) -> Path:
    """Compile the extension in-tree (g++ for the host builder, hipcc --offload-arch=gfx950)."""
    import subprocess
    cmd = ["make", "-C", str(_HERE / "csrc"), "all"]
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def _preload_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own libamdhip64.so with the
    same SONAME (libamdhip64.so.7) as /opt/rocm's; whichever is mapped first satisfies
    libnewmap_amd.so's DT_NEEDED.  If this library came first and torch were imported later, torch
    would map a SECOND runtime that sees no GPU.  So when torch is installed, map its runtime
    before ours (without importing torch).  NEWMAP_AMD_HIP_RUNTIME=system opts out."""
    if os.environ.get("NEWMAP_AMD_HIP_RUNTIME", "auto") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise EngineMissingError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C newmap_amd/csrc`).  newmap_amd has no CPU fallback.")
    _preload_hip_runtime()
    L = ctypes.CDLL(str(LIB_PATH))
    c = ctypes
    vp, u64, u32, i32, u8 = c.c_void_p, c.c_uint64, c.c_uint32, c.c_int, c.c_uint8
    pp = c.POINTER(c.c_void_p)
    L.nm_last_error.restype = c.c_char_p
    L.nm_version.restype = c.c_char_p
    L.nm_index_build.restype = i32
    L.nm_index_build.argtypes = [c.c_char_p, c.c_char_p, u8, u8]
    L.nm_index_build_device.restype = i32
    L.nm_index_build_device.argtypes = [c.c_char_p, c.c_char_p, u8, u8, i32]
    L.nm_index_open.restype = i32
    L.nm_index_open.argtypes = [c.c_char_p, i32, i32, pp]
    L.nm_dev_free_bytes.restype = u64
    L.nm_dev_free_bytes.argtypes = [i32]
    L.nm_index_open_budget.restype = i32
    L.nm_index_open_budget.argtypes = [c.c_char_p, i32, i32, u64, pp]
    L.nm_index_close.restype = None
    L.nm_index_close.argtypes = [vp]
    L.nm_index_info.restype = u64
    L.nm_index_info.argtypes = [vp, i32]
    L.nm_count_kmers.restype = i32
    L.nm_count_kmers.argtypes = [vp, vp, vp, u64, vp]
    L.nm_count_from_sequence.restype = i32
    L.nm_count_from_sequence.argtypes = [vp, vp, u64, vp, vp, u64, vp]
    L.nm_min_unique_segment.restype = i32
    L.nm_min_unique_segment.argtypes = [vp, vp, u64, u64, u32, u32, u32, i32, i32, vp, vp, vp]
    L.nm_fixed_k_segment.restype = i32
    L.nm_fixed_k_segment.argtypes = [vp, vp, u64, u64, vp, u32, i32, i32, vp, vp, vp]
    L.nm_upper_bound_segment.restype = i32
    L.nm_upper_bound_segment.argtypes = [vp, vp, u64, u64, u32, vp]
    L.nm_min_unique_segment_dev.restype = i32
    L.nm_min_unique_segment_dev.argtypes = [vp, vp, u64, u64, u32, u32, i32, i32, vp, vp, vp]
    L.nm_fixed_k_segment_dev.restype = i32
    L.nm_fixed_k_segment_dev.argtypes = [vp, vp, u64, u64, vp, u32, i32, i32, vp, vp, vp]
    L.nm_set_option.restype = i32
    L.nm_set_option.argtypes = [vp, i32, c.c_int64]
    L.nm_dev_alloc.restype = i32
    L.nm_dev_alloc.argtypes = [i32, u64, pp]
    L.nm_dev_free.restype = i32
    L.nm_dev_free.argtypes = [i32, vp]
    L.nm_dev_upload.restype = i32
    L.nm_dev_upload.argtypes = [i32, vp, vp, u64]
    L.nm_dev_download.restype = i32
    L.nm_dev_download.argtypes = [i32, vp, vp, u64]
    L.nm_dev_sync.restype = i32
    L.nm_dev_sync.argtypes = [i32]
    L.nm_device_count.restype = i32
    L.nm_search_fasta.restype = i32
    L.nm_search_fasta.argtypes = [vp, c.c_char_p, c.c_char_p, vp, u32, i32, i32, u64,
                                  c.POINTER(c.c_char_p), u32, c.POINTER(c.c_char_p), u32,
                                  RECORD_CALLBACK, vp, c.POINTER(SearchSummary)]
    L.nm_search_fasta_shard.restype = i32
    L.nm_search_fasta_shard.argtypes = L.nm_search_fasta.argtypes + [i32, i32]
    L.nm_search_fasta_shard_ex.restype = i32
    L.nm_search_fasta_shard_ex.argtypes = L.nm_search_fasta_shard.argtypes + [vp, u64, c.POINTER(u64)]
    L.nm_guard_fasta.restype = i32
    L.nm_guard_fasta.argtypes = [vp, c.c_char_p, vp, u32, i32, i32, u64, c.POINTER(c.c_char_p), u32, c.POINTER(c.c_char_p), u32, vp, u64]
    L.nm_stream_release.restype = i32
    L.nm_stream_release.argtypes = [vp, vp]
    L.nm_search_segment_multi.restype = i32
    L.nm_search_segment_multi.argtypes = [vp, u32, vp, u32, u64, u64, vp, u32, i32, i32, i32, vp, vp, vp]
    L.nm_track_file.restype = i32
    L.nm_track_file.argtypes = [i32, c.c_char_p, c.c_char_p, i32, u32, c.c_char_p, c.c_char_p,
                                c.POINTER(u64), c.POINTER(u64)]
    L.nm_timing_read.restype = i32
    L.nm_timing_read.argtypes = [vp, c.POINTER(u64), c.POINTER(c.c_double), c.POINTER(c.c_double)]
    L.nm_timing_read_kind.restype = i32
    L.nm_timing_read_kind.argtypes = [vp, i32, c.POINTER(u64), c.POINTER(c.c_double), c.POINTER(c.c_double)]
    L.nm_index_has_record.restype = i32
    L.nm_index_has_record.argtypes = [vp, u64, u64]
    L.nm_index_records.restype = u64
    L.nm_index_records.argtypes = [vp, vp, vp, u64]
    L.nm_fingerprint_join.restype = u64
    L.nm_fingerprint_join.argtypes = [u64, u64, u64]
    L.nm_fingerprint_sequence.restype = u64
    L.nm_fingerprint_sequence.argtypes = [vp, u64]
    L.nm_guard_segment_dev.restype = i32
    L.nm_guard_segment_dev.argtypes = [vp, vp, u64, u64, vp, u32, i32, u32, i32, vp, vp]
    L.nm_guard_segment.restype = i32
    L.nm_guard_segment.argtypes = [vp, vp, u64, u64, vp, u32, i32, u32, i32, vp]
    _lib = L
    return L


def last_error() -> str:
    return lib().nm_last_error().decode("utf-8", "replace")


def raise_for(code: int, not_found_detail: str | None = None):
    """Map a C-ABI return code to the exception class the reference raises for the same
    condition (src/newmap-generate-index.c:32-54, src/newmap-count.c:13-16,
    newmap/search.py:719-722)."""
    if code == NM_OK:
        return
    msg = last_error()
    if code == NM_E_FILE_OPEN:
        raise FileNotFoundError(msg)
    if code == NM_E_ALLOC:
        raise MemoryError(msg)
    if code == NM_E_FILE_EXISTS:
        raise FileExistsError(msg)
    if code in (NM_E_FILE_WRITE, NM_E_FILE_FORMAT):
        raise OSError(msg)
    if code == NM_E_ARGUMENT:
        raise ValueError(msg)
    if code == NM_E_KMER_NOT_FOUND:
        raise RuntimeError(not_found_detail or msg)
    if code == NM_E_TOO_LARGE:
        raise OverflowError(msg)
    raise RuntimeError(msg)

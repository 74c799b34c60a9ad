// nm_track.hip -- `newmap track` on the device.
//
// Reference: newmap/track.py:22-121.  For one `<chr>.unique.uintN` array u[0..n) and a read length k:
//   marks[i]   = 0 < u[i] <= k                       (a uniquely mappable k-window starts at i)
//   covered[i] = sum of marks[i-k+1 .. i]            (track.py:46-52: +1/-1 marks, cumulative sum)
//   multi-read mappability = covered / k  (WIG, one formatted value per line, track.py:96-121)
//   single-read mappability = covered > 0 (BED, one line per run of equal values, track.py:62-93)
// All of it is scans and a stream compaction, HBM-stream bound:
//   k_track_marks   u -> marks (u32 0/1)                      + hierarchical exclusive scan
//   k_track_cover   prefix sums -> covered, run-boundary flags, WIG line length per base
//   (scan of the flags -> run starts; scan of the line lengths -> byte offsets)
//   k_track_runs    compaction of the run starts
//   k_track_wig     scatter of the formatted lines (k+1 distinct strings, formatted on the host by
//                   printf, which rounds like the reference's f"{v:.{d}f}")
// The host appends the BED lines (a handful per Mbp) and the WIG text to the output files.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/newmap_amd.h"
#include "nm_internal.h"

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            nm_set_error("HIP error %d (%s) at %s:%d: %s", (int)e__, hipGetErrorString(e__),  \
                         __FILE__, __LINE__, #expr);                                          \
            return NM_E_DEVICE;                                                               \
        }                                                                                     \
    } while (0)

#include "nm_scan.hip.h"

// ---------------------------------------------------------------- track kernels --------------
__global__ __launch_bounds__(TB) void k_track_marks(const void *__restrict__ u, int elem_bytes, uint64_t n, uint32_t k,
                                                    uint64_t *__restrict__ marks) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = elem_bytes == 1 ? ((const uint8_t *)u)[i] : (elem_bytes == 2 ? ((const uint16_t *)u)[i] : ((const uint32_t *)u)[i]);
    marks[i] = (v != 0 && v <= k) ? 1u : 0u;
}

// prefix[i] = marks[0..i) (exclusive); covered[i] = prefix[i+1] - prefix[max(i+1-k, 0)]
__global__ __launch_bounds__(TB) void k_track_cover(const uint64_t *__restrict__ prefix, uint64_t total, uint64_t n, uint32_t k,
                                                    uint32_t *__restrict__ covered, uint64_t *__restrict__ run_flag,
                                                    uint64_t *__restrict__ line_len, const uint8_t *__restrict__ len_table) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    auto cov = [&](uint64_t j) -> uint32_t {
        const uint64_t hi = j + 1 < n ? prefix[j + 1] : total;
        const uint64_t lo = j + 1 >= k ? prefix[j + 1 - k] : 0;
        return (uint32_t)(hi - lo);
    };
    const uint32_t c = cov(i);
    covered[i] = c;
    const bool cur = c > 0;
    const bool prev = i ? cov(i - 1) > 0 : !cur;          // position 0 always opens a run
    run_flag[i] = (i == 0 || cur != prev) ? 1u : 0u;
    if (line_len) line_len[i] = len_table[c];
}

__global__ __launch_bounds__(TB) void k_track_runs(const uint64_t *__restrict__ flag_scan, const uint32_t *__restrict__ covered,
                                                   uint64_t n, uint64_t n_runs, uint64_t *__restrict__ run_start,
                                                   uint8_t *__restrict__ run_value) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint64_t here = flag_scan[i], next = i + 1 < n ? flag_scan[i + 1] : n_runs;
    if (next != here) { run_start[here] = i; run_value[here] = covered[i] > 0; }
}

__global__ __launch_bounds__(TB) void k_track_wig(const uint64_t *__restrict__ offsets, const uint32_t *__restrict__ covered,
                                                  uint64_t n, const uint8_t *__restrict__ text_table, uint32_t stride,
                                                  const uint8_t *__restrict__ len_table, uint8_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = covered[i];
    const uint8_t *src = text_table + (uint64_t)c * stride;
    uint8_t *dst = out + offsets[i];
    const uint32_t len = len_table[c];
    for (uint32_t j = 0; j < len; j++) dst[j] = src[j];
}

// ---------------------------------------------------------------- host ---------------------------
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(uint64_t bytes) {
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 8));
        return NM_OK;
    }
};
}  // namespace

extern "C" int nm_track_file(int device, const char *unique_path, const char *chr_name, int elem_bytes, uint32_t k,
                             const char *bed_path, const char *wig_path, uint64_t *n_positions, uint64_t *n_runs_out) {
    if (!unique_path || !chr_name || k == 0) { nm_set_error("bad argument"); return NM_E_ARGUMENT; }
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4) { nm_set_error("elem_bytes must be 1, 2 or 4"); return NM_E_ARGUMENT; }
    if (device < 0) { nm_set_error("device %d: there is no CPU path", device); return NM_E_DEVICE; }
    FILE *fp = fopen(unique_path, "rb");
    if (!fp) { nm_set_error("Unique count file does not exist: %s", unique_path); return NM_E_FILE_OPEN; }
    fseeko(fp, 0, SEEK_END);
    const uint64_t bytes = (uint64_t)ftello(fp);
    fseeko(fp, 0, SEEK_SET);
    const uint64_t n = bytes / (uint64_t)elem_bytes;
    std::vector<uint8_t> host(bytes ? bytes : 1);
    if (bytes && fread(host.data(), 1, bytes, fp) != bytes) { fclose(fp); nm_set_error("short read on %s", unique_path); return NM_E_FILE_OPEN; }
    fclose(fp);
    if (n_positions) *n_positions = n;
    if (n_runs_out) *n_runs_out = 0;

    // the k+1 distinct WIG lines (track.py:114-121 float_format, decimals = ceil(log10 k), :224)
    const int decimals = (int)std::ceil(std::log10((double)k));
    const uint32_t stride = 32;
    std::vector<uint8_t> text((size_t)(k + 1) * stride, 0), lens(k + 1, 0);
    for (uint32_t c = 0; c <= k; c++) {
        char line[64];
        int len = c == 0 ? snprintf(line, sizeof line, "0.0\n")
                         : snprintf(line, sizeof line, "%.*f\n", decimals, (double)c / (double)k);
        if (len <= 0 || len > (int)stride) { nm_set_error("value formatting failed"); return NM_E_ARGUMENT; }
        memcpy(&text[(size_t)c * stride], line, (size_t)len);
        lens[c] = (uint8_t)len;
    }

    FILE *bed = nullptr, *wig = nullptr;
    if (bed_path && !(bed = fopen(bed_path, "ab"))) { nm_set_error("could not open %s: %s", bed_path, strerror(errno)); return NM_E_FILE_WRITE; }
    if (wig_path && !(wig = fopen(wig_path, "ab"))) { if (bed) fclose(bed); nm_set_error("could not open %s: %s", wig_path, strerror(errno)); return NM_E_FILE_WRITE; }
    auto finish = [&](int rc) { if (bed) fclose(bed); if (wig) fclose(wig); return rc; };
    if (wig) fprintf(wig, "fixedStep chrom=%s start=1 step=1 span=1\n", chr_name);       // track.py:17-18,101-104
    if (n == 0) return finish(NM_OK);

    HIP_TRY(hipSetDevice(device));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreate(&st));
    DevBuf d_u, d_a, d_b, d_cov, d_scratch, d_total, d_text, d_lens, d_runs, d_runv, d_out;
    int rc;
    const uint64_t scratch_n = n / TILE + n / ((uint64_t)TILE * TILE) + 8192;
    if ((rc = d_u.alloc(bytes)) || (rc = d_a.alloc(n * 8)) || (rc = d_b.alloc(n * 8)) || (rc = d_cov.alloc(n * 4)) ||
        (rc = d_scratch.alloc(scratch_n * 8)) || (rc = d_total.alloc(16)) || (rc = d_text.alloc(text.size())) ||
        (rc = d_lens.alloc(lens.size()))) { (void)hipStreamDestroy(st); return finish(rc); }
    auto fail = [&](int code) { (void)hipStreamDestroy(st); return finish(code); };
#define TRY_RC(call) do { int r__ = (call); if (r__ != NM_OK) return fail(r__); } while (0)
#define TRY_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { nm_set_error("HIP error %d (%s): %s", (int)e__, hipGetErrorString(e__), #expr); return fail(NM_E_DEVICE); } } while (0)
    TRY_HIP(hipMemcpyAsync(d_u.p, host.data(), bytes, hipMemcpyHostToDevice, st));
    TRY_HIP(hipMemcpyAsync(d_text.p, text.data(), text.size(), hipMemcpyHostToDevice, st));
    TRY_HIP(hipMemcpyAsync(d_lens.p, lens.data(), lens.size(), hipMemcpyHostToDevice, st));
    const unsigned grid = (unsigned)((n + TB - 1) / TB);
    uint64_t *A = (uint64_t *)d_a.p, *B = (uint64_t *)d_b.p, *total = (uint64_t *)d_total.p;
    uint32_t *cov = (uint32_t *)d_cov.p;
    // marks -> prefix sums (in A)
    hipLaunchKernelGGL(k_track_marks, dim3(grid), dim3(TB), 0, st, d_u.p, elem_bytes, n, k, A);
    TRY_RC(scan_exclusive(A, n, (uint64_t *)d_scratch.p, total, st));
    uint64_t h_total = 0;
    TRY_HIP(hipMemcpyAsync(&h_total, total, 8, hipMemcpyDeviceToHost, st));
    TRY_HIP(hipStreamSynchronize(st));
    // covered, run flags (in B), line lengths (reuse A after cover has read it: write to a third array)
    DevBuf d_len64;
    if (wig) TRY_RC(d_len64.alloc(n * 8));
    hipLaunchKernelGGL(k_track_cover, dim3(grid), dim3(TB), 0, st, A, h_total, n, k, cov, B,
                       wig ? (uint64_t *)d_len64.p : (uint64_t *)nullptr, (const uint8_t *)d_lens.p);
    // run starts
    TRY_RC(scan_exclusive(B, n, (uint64_t *)d_scratch.p, total, st));
    uint64_t n_runs = 0;
    TRY_HIP(hipMemcpyAsync(&n_runs, total, 8, hipMemcpyDeviceToHost, st));
    TRY_HIP(hipStreamSynchronize(st));
    if (n_runs_out) *n_runs_out = n_runs;
    if (bed) {
        TRY_RC(d_runs.alloc(n_runs * 8));
        TRY_RC(d_runv.alloc(n_runs));
        hipLaunchKernelGGL(k_track_runs, dim3(grid), dim3(TB), 0, st, B, cov, n, n_runs, (uint64_t *)d_runs.p, (uint8_t *)d_runv.p);
        std::vector<uint64_t> starts(n_runs);
        std::vector<uint8_t> vals(n_runs);
        TRY_HIP(hipMemcpyAsync(starts.data(), d_runs.p, n_runs * 8, hipMemcpyDeviceToHost, st));
        TRY_HIP(hipMemcpyAsync(vals.data(), d_runv.p, n_runs, hipMemcpyDeviceToHost, st));
        TRY_HIP(hipStreamSynchronize(st));
        for (uint64_t r = 0; r < n_runs; r++) {           // track.py:82-93 line format
            const uint64_t end = r + 1 < n_runs ? starts[r + 1] : n;
            fprintf(bed, "%s\t%llu\t%llu\tk%u\t%u\t.\n", chr_name, (unsigned long long)starts[r], (unsigned long long)end, k, (unsigned)vals[r]);
        }
    }
    if (wig) {
        uint64_t *L = (uint64_t *)d_len64.p;
        TRY_RC(scan_exclusive(L, n, (uint64_t *)d_scratch.p, total, st));
        uint64_t out_bytes = 0;
        TRY_HIP(hipMemcpyAsync(&out_bytes, total, 8, hipMemcpyDeviceToHost, st));
        TRY_HIP(hipStreamSynchronize(st));
        TRY_RC(d_out.alloc(out_bytes));
        hipLaunchKernelGGL(k_track_wig, dim3(grid), dim3(TB), 0, st, L, cov, n, (const uint8_t *)d_text.p, stride,
                           (const uint8_t *)d_lens.p, (uint8_t *)d_out.p);
        std::vector<uint8_t> out(out_bytes);
        TRY_HIP(hipMemcpyAsync(out.data(), d_out.p, out_bytes, hipMemcpyDeviceToHost, st));
        TRY_HIP(hipStreamSynchronize(st));
        if (fwrite(out.data(), 1, out_bytes, wig) != out_bytes) { nm_set_error("could not write %s", wig_path); return fail(NM_E_FILE_WRITE); }
    }
    TRY_HIP(hipGetLastError());
    (void)hipStreamDestroy(st);
    return finish(NM_OK);
}

// nm_build_device.hip -- index construction with the suffix sort on the MI355X.
//
// Same pipeline as nm_build.cpp (FASTA -> both-strand text -> suffix array -> BWT -> blocks -> file)
// with the expensive stage, the suffix array, computed on the device by prefix doubling
// (Manber & Myers 1993) on top of rocPRIM's device radix sort:
//
//   key[i] = first 21 symbols of suffix i (63 bits), sort (key, i)                        h = 21
//   repeat: head[j] = first sorted position of j's group (max-scan of group starts)
//           rank[SA[j]] = head[j];  key[j] = head[j] << 32 | rank[SA[j] + h];  sort;  h *= 2
//   until every group has one member.
//
// Every pass is a full-width radix sort of n (key, position) pairs at HBM speed; uniform DNA needs
// one refinement pass, 50 kb tandem arrays about a dozen.  The result is the true lexicographic
// suffix array, so the index file is byte-identical to the host builder's (tests/test_gpu_parity.py).
// Limit: texts below 2^31 symbols (genomes up to ~1.07 Gbp); larger inputs use the host sorter.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

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

#define SA_BLOCK 256
#define SA_H0 21u

__global__ __launch_bounds__(SA_BLOCK) void k_sa_init(const uint8_t *__restrict__ T, uint32_t n, uint64_t *__restrict__ key,
                                                      uint32_t *__restrict__ idx) {
    const uint64_t i = blockIdx.x * (uint64_t)SA_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t k = 0;
#pragma unroll
    for (uint32_t j = 0; j < SA_H0; j++) {
        const uint64_t p = i + j;
        k |= (uint64_t)(p < n ? T[p] : 0) << (3 * (SA_H0 - 1 - j));
    }
    key[i] = k;
    idx[i] = (uint32_t)i;
}

// start[j] = j where a new group begins, else 0 (for the max-scan); flag[j] = 1 at group starts
__global__ __launch_bounds__(SA_BLOCK) void k_sa_flags(const uint64_t *__restrict__ key, uint32_t n, uint32_t *__restrict__ start,
                                                       uint32_t *__restrict__ flag) {
    const uint64_t j = blockIdx.x * (uint64_t)SA_BLOCK + threadIdx.x;
    if (j >= n) return;
    const bool head = j == 0 || key[j] != key[j - 1];
    start[j] = head ? (uint32_t)j : 0u;
    flag[j] = head ? 1u : 0u;
}

__global__ __launch_bounds__(SA_BLOCK) void k_sa_scatter(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ head, uint32_t n,
                                                         uint32_t *__restrict__ rank) {
    const uint64_t j = blockIdx.x * (uint64_t)SA_BLOCK + threadIdx.x;
    if (j < n) rank[idx[j]] = head[j];
}

__global__ __launch_bounds__(SA_BLOCK) void k_sa_keys(const uint32_t *__restrict__ idx, const uint32_t *__restrict__ head,
                                                      const uint32_t *__restrict__ rank, uint32_t n, uint64_t h,
                                                      uint64_t *__restrict__ key) {
    const uint64_t j = blockIdx.x * (uint64_t)SA_BLOCK + threadIdx.x;
    if (j >= n) return;
    const uint64_t p = (uint64_t)idx[j] + h;
    // a member of a group that is still unresolved has not reached the terminator, so p < n for it
    key[j] = ((uint64_t)head[j] << 32) | (p < n ? rank[p] : 0u);
}

namespace {
struct DBuf {
    void *p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
};

int dev_alloc(DBuf &b, uint64_t bytes) {
    HIP_TRY(hipMalloc(&b.p, bytes ? bytes : 8));
    return NM_OK;
}

struct Ctx { int device; };

int device_suffix_array(const uint8_t *T, uint64_t n64, int32_t *SA, void *ctx_) {
    const Ctx *ctx = (const Ctx *)ctx_;
    if (n64 >= (1ULL << 31)) { nm_set_error("device suffix sort handles texts below 2^31 symbols"); return NM_E_TOO_LARGE; }
    const uint32_t n = (uint32_t)n64;
    const bool verbose = getenv("NEWMAP_AMD_VERBOSE") && *getenv("NEWMAP_AMD_VERBOSE") != '0';
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreate(&st));
    DBuf dT, dK0, dK1, dI0, dI1, dStart, dFlag, dRank, dTmp, dCount;
    int rc;
    if ((rc = dev_alloc(dT, n)) || (rc = dev_alloc(dK0, (uint64_t)n * 8)) || (rc = dev_alloc(dK1, (uint64_t)n * 8)) ||
        (rc = dev_alloc(dI0, (uint64_t)n * 4)) || (rc = dev_alloc(dI1, (uint64_t)n * 4)) || (rc = dev_alloc(dStart, (uint64_t)n * 4)) ||
        (rc = dev_alloc(dFlag, (uint64_t)n * 4)) || (rc = dev_alloc(dRank, (uint64_t)n * 4)) || (rc = dev_alloc(dCount, 16))) {
        (void)hipStreamDestroy(st);
        return rc;
    }
    auto fail = [&](int code) { (void)hipStreamDestroy(st); return code; };
#define TRYH(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { nm_set_error("HIP error %d (%s): %s", (int)e__, hipGetErrorString(e__), #expr); return fail(NM_E_DEVICE); } } while (0)
    TRYH(hipMemcpyAsync(dT.p, T, n, hipMemcpyHostToDevice, st));
    rocprim::double_buffer<uint64_t> keys((uint64_t *)dK0.p, (uint64_t *)dK1.p);
    rocprim::double_buffer<uint32_t> vals((uint32_t *)dI0.p, (uint32_t *)dI1.p);
    uint32_t *start = (uint32_t *)dStart.p, *flag = (uint32_t *)dFlag.p, *rank = (uint32_t *)dRank.p;
    uint32_t *count = (uint32_t *)dCount.p;
    // temporary storage: the largest need of the three primitives
    size_t t_sort = 0, t_scan = 0, t_red = 0;
    TRYH(rocprim::radix_sort_pairs(nullptr, t_sort, keys, vals, n, 0, 64, st));
    TRYH(rocprim::inclusive_scan(nullptr, t_scan, start, start, n, rocprim::maximum<uint32_t>(), st));
    TRYH(rocprim::reduce(nullptr, t_red, flag, count, 0u, n, rocprim::plus<uint32_t>(), st));
    size_t t_bytes = t_sort > t_scan ? t_sort : t_scan;
    if (t_red > t_bytes) t_bytes = t_red;
    if ((rc = dev_alloc(dTmp, t_bytes)) != NM_OK) return fail(rc);

    const unsigned grid = (unsigned)(((uint64_t)n + SA_BLOCK - 1) / SA_BLOCK);
    hipLaunchKernelGGL(k_sa_init, dim3(grid), dim3(SA_BLOCK), 0, st, (const uint8_t *)dT.p, n, keys.current(), vals.current());
    size_t tb = t_bytes;
    TRYH(rocprim::radix_sort_pairs(dTmp.p, tb, keys, vals, n, 0, 63, st));
    unsigned key_bits = 33;
    for (uint32_t m = n; m > 1; m >>= 1) key_bits++;            // head (high word) needs ceil(log2 n) bits
    if (key_bits > 64) key_bits = 64;
    int rounds = 0;
    for (uint64_t h = SA_H0;; h *= 2) {
        hipLaunchKernelGGL(k_sa_flags, dim3(grid), dim3(SA_BLOCK), 0, st, (const uint64_t *)keys.current(), n, start, flag);
        tb = t_bytes;
        TRYH(rocprim::inclusive_scan(dTmp.p, tb, start, start, n, rocprim::maximum<uint32_t>(), st));   // head of each position
        tb = t_bytes;
        TRYH(rocprim::reduce(dTmp.p, tb, flag, count, 0u, n, rocprim::plus<uint32_t>(), st));
        uint32_t groups = 0;
        TRYH(hipMemcpyAsync(&groups, count, 4, hipMemcpyDeviceToHost, st));
        TRYH(hipStreamSynchronize(st));
        if (verbose) fprintf(stderr, "[device sa] h=%llu: %u groups of %u suffixes\n", (unsigned long long)h, groups, n);
        if (groups == n) break;
        if (h > 2ULL * n + SA_H0 || ++rounds > 40) { nm_set_error("device suffix sort did not converge"); return fail(NM_E_DEVICE); }
        hipLaunchKernelGGL(k_sa_scatter, dim3(grid), dim3(SA_BLOCK), 0, st, (const uint32_t *)vals.current(), (const uint32_t *)start, n, rank);
        hipLaunchKernelGGL(k_sa_keys, dim3(grid), dim3(SA_BLOCK), 0, st, (const uint32_t *)vals.current(), (const uint32_t *)start,
                           (const uint32_t *)rank, n, h, keys.current());
        tb = t_bytes;
        TRYH(rocprim::radix_sort_pairs(dTmp.p, tb, keys, vals, n, 0, key_bits, st));
    }
    TRYH(hipMemcpyAsync(SA, vals.current(), (uint64_t)n * 4, hipMemcpyDeviceToHost, st));
    TRYH(hipStreamSynchronize(st));
    TRYH(hipGetLastError());
    (void)hipStreamDestroy(st);
    return NM_OK;
#undef TRYH
}
}  // namespace

extern "C" int nm_index_build_device(const char *fasta_path, const char *index_path, uint8_t sa_ratio, uint8_t seed_len, int device) {
    if (device < 0) { nm_set_error("device %d: the device builder needs a GPU (nm_index_build is the host builder)", device); return NM_E_DEVICE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) { nm_set_error("device %d requested but %d HIP device(s) are visible", device, ndev); return NM_E_DEVICE; }
    Ctx ctx{device};
    return nm_index_build_impl(fasta_path, index_path, sa_ratio, seed_len, device_suffix_array, &ctx);
}

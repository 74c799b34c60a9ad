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
// This full-width variant takes texts below 2^31 symbols (genomes up to ~1.07 Gbp); larger ones go through
// device_bwt_large further down (64-bit, bucketed, refinement of the tied groups only).
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/newmap_amd.h"
#include "nm_format.h"
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
// LCP bytes (nm_format.h: off_lcp): lcp[j] = bases the suffixes of rows j - 1 and j share, capped; a separator (symbols below 2)
// matches nothing.  One lane per row, the text and the finished suffix array resident.
template <class IDX>
__global__ __launch_bounds__(SA_BLOCK) void k_lcp_bytes(const uint8_t *__restrict__ T, const IDX *__restrict__ SA, unsigned long long first,
                                                        unsigned long long n, uint8_t *__restrict__ lcp) {
    const unsigned long long j = first + blockIdx.x * (unsigned long long)SA_BLOCK + threadIdx.x;
    if (j > n) return;
    if (j == 0 || j == n) { lcp[j] = 0; return; }
    const unsigned long long p = (unsigned long long)SA[j], q = (unsigned long long)SA[j - 1];
    unsigned h = 0;
    while (h < NM_LCP_CAP) {
        const uint8_t a = T[p + h];
        if (a < 2 || a != T[q + h]) break;                 // (the text ends with symbol 0: the loop stops inside it)
        h++;
    }
    lcp[j] = (uint8_t)h;
}

struct DBuf {
    void *p = nullptr;
    ~DBuf() { reset(); }
    void reset() { if (p) (void)hipFree(p); p = nullptr; }
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
    if (uint8_t *lcp_host = nm_build_lcp_buffer()) {        // the LCP bytes while text and suffix array are here (dK0 is free now: n + 1 bytes of it)
        uint8_t *d_lcp = (uint8_t *)(keys.current() == (uint64_t *)dK0.p ? dK1.p : dK0.p);
        hipLaunchKernelGGL((k_lcp_bytes<uint32_t>), dim3((unsigned)(((uint64_t)n + 1 + SA_BLOCK - 1) / SA_BLOCK)), dim3(SA_BLOCK), 0, st, (const uint8_t *)dT.p,
                           (const uint32_t *)vals.current(), 0ULL, (unsigned long long)n, d_lcp);
        TRYH(hipMemcpyAsync(lcp_host, d_lcp, (uint64_t)n + 1, hipMemcpyDeviceToHost, st));
        TRYH(hipStreamSynchronize(st));
        nm_build_lcp_done();
    }
    TRYH(hipStreamSynchronize(st));
    TRYH(hipGetLastError());
    (void)hipStreamDestroy(st);
    return NM_OK;
#undef TRYH
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// Large texts (2^31 symbols and more: the 3 Gbp north-star genome has 6.2 G), everything in 64 bits.
// The full-width sort of the 32-bit path would need 52 bytes per symbol; this one keeps only the suffix
// array and the rank array (8 + 8 bytes per symbol, 99 GB for a human genome) resident:
//   A. initial order by the first 21 symbols, one bucket of leading symbols at a time: collect the
//      bucket's positions, sort (key, position), append to SA, rank[position] = first sorted index of its
//      key group;
//   B. refinement of the groups that are still tied, and only of them (prefix doubling with discarding,
//      Larsson & Sadakane 2007): the list of tied sorted positions is sorted by (group, rank[SA + h]) with
//      two stable radix sorts, written back, re-grouped, and the members that became singletons leave the list;
//   C. bw[j] = T[SA[j] - 1] | strand flag is gathered on the device, so n bytes come back, not 8 n.
typedef unsigned long long u64;

// bucket sizes: grid-stride, counted in LDS first (at most 4096 buckets), one global atomic per block and bucket
#define SA_MAX_BUCKET_BITS 12u
__global__ __launch_bounds__(SA_BLOCK) void k_big_hist(const uint8_t *__restrict__ T, u64 n, unsigned bits, u64 *__restrict__ hist) {
    __shared__ unsigned bins[1u << SA_MAX_BUCKET_BITS];
    const unsigned nb = 1u << bits;
    for (unsigned b = threadIdx.x; b < nb; b += SA_BLOCK) bins[b] = 0;
    __syncthreads();
    const u64 per_block = (n + gridDim.x - 1) / gridDim.x;
    const u64 a = blockIdx.x * per_block, e = a + per_block < n ? a + per_block : n;
    for (u64 i = a + threadIdx.x; i < e; i += SA_BLOCK) {
        unsigned b = 0;
        for (unsigned j = 0; j < bits / 3; j++) b = (b << 3) | (unsigned)(i + j < n ? T[i + j] : 0);
        atomicAdd(&bins[b], 1u);
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < nb; b += SA_BLOCK)
        if (bins[b]) atomicAdd(&hist[b], (u64)bins[b]);
}

// positions whose leading symbols spell bucket b, in any order (they are sorted next).  A block owns a
// contiguous chunk of the text: it counts its members, reserves their slots with ONE atomic and writes them on a
// second sweep (one atomic per wave on a single counter made this kernel 90 % of the build at 6.2 G symbols).
__global__ __launch_bounds__(SA_BLOCK) void k_big_collect(const uint8_t *__restrict__ T, u64 n, unsigned bits, u64 bucket,
                                                          u64 *__restrict__ idx, u64 *__restrict__ key, u64 *__restrict__ counter) {
    __shared__ unsigned long long cursor;
    __shared__ unsigned total;
    const u64 per_block = (n + gridDim.x - 1) / gridDim.x;
    const u64 a = blockIdx.x * per_block, e = a + per_block < n ? a + per_block : n;
    const unsigned lane = threadIdx.x & 63;
    auto member = [&](u64 i) -> bool {
        if (i >= e) return false;
        u64 b = 0;
        for (unsigned j = 0; j < bits / 3; j++) b = (b << 3) | (u64)(i + j < n ? T[i + j] : 0);
        return b == bucket;
    };
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    unsigned mine = 0;
    for (u64 i = a + threadIdx.x; i < e; i += SA_BLOCK) mine += member(i) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if (lane == 0 && mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) cursor = total ? atomicAdd(counter, (u64)total) : 0ULL;
    __syncthreads();
    if (!total) return;
    for (u64 base_i = a; base_i < e; base_i += SA_BLOCK) {     // all lanes take every trip: the ballots need whole waves
        const u64 i = base_i + threadIdx.x;
        const bool m = member(i);
        const u64 mask = __ballot(m);
        if (!mask) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&cursor, (unsigned long long)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (m) {
            const u64 at = base + (u64)__popcll(mask & ((1ULL << lane) - 1ULL));
            u64 k = 0;
#pragma unroll
            for (uint32_t j = 0; j < SA_H0; j++) k |= (u64)(i + j < n ? T[i + j] : 0) << (3 * (SA_H0 - 1 - j));
            idx[at] = i;
            key[at] = k;
        }
    }
}

// start[k] = global sorted index of k if a new key group starts there, else 0 (max-scanned into the group head)
__global__ __launch_bounds__(SA_BLOCK) void k_big_starts(const u64 *__restrict__ key, u64 m, u64 offset, u64 *__restrict__ start) {
    const u64 k = blockIdx.x * (u64)SA_BLOCK + threadIdx.x;
    if (k < m) start[k] = (k == 0 || key[k] != key[k - 1]) ? offset + k : 0ULL;
}

__global__ __launch_bounds__(SA_BLOCK) void k_big_place(const u64 *__restrict__ idx, const u64 *__restrict__ head, u64 m, u64 offset,
                                                        u64 *__restrict__ SA, u64 *__restrict__ rank) {
    const u64 k = blockIdx.x * (u64)SA_BLOCK + threadIdx.x;
    if (k < m) { SA[offset + k] = idx[k]; rank[idx[k]] = head[k]; }
}

// 1 where sorted position j belongs to a group of more than one suffix
__global__ __launch_bounds__(SA_BLOCK) void k_big_tied(const u64 *__restrict__ SA, const u64 *__restrict__ rank, u64 first, u64 n, uint8_t *__restrict__ flag) {
    const u64 j = first + blockIdx.x * (u64)SA_BLOCK + threadIdx.x;
    if (j >= n) return;
    const bool head = rank[SA[j]] == j;
    const bool next_head = j + 1 >= n || rank[SA[j + 1]] == j + 1;
    flag[j] = (head && next_head) ? 0 : 1;
}

__global__ __launch_bounds__(SA_BLOCK) void k_big_gather(const u64 *__restrict__ list, u64 m, const u64 *__restrict__ SA, const u64 *__restrict__ rank,
                                                         u64 n, u64 h, u64 *__restrict__ sa, u64 *__restrict__ head, u64 *__restrict__ key2,
                                                         uint32_t *__restrict__ perm) {
    const u64 k = blockIdx.x * (u64)SA_BLOCK + threadIdx.x;
    if (k >= m) return;
    const u64 p = SA[list[k]];
    sa[k] = p;
    head[k] = rank[p];
    key2[k] = p + h < n ? rank[p + h] : 0ULL;              // a tied suffix has not reached the terminator, so p + h < n
    perm[k] = (uint32_t)k;
}

__global__ __launch_bounds__(SA_BLOCK) void k_big_pick(const uint32_t *__restrict__ perm, const u64 *__restrict__ src, u64 m, u64 *__restrict__ dst) {
    const u64 k = blockIdx.x * (u64)SA_BLOCK + threadIdx.x;
    if (k < m) dst[k] = src[perm[k]];
}

// after the two sorts perm[k] names the member that belongs at list slot k: new group starts
__global__ __launch_bounds__(SA_BLOCK) void k_big_regroup(const uint32_t *__restrict__ perm, const u64 *__restrict__ head, const u64 *__restrict__ key2,
                                                          const u64 *__restrict__ list, u64 m, u64 *__restrict__ start) {
    const u64 k = blockIdx.x * (u64)SA_BLOCK + threadIdx.x;
    if (k >= m) return;
    bool first = k == 0;
    if (!first) {
        const uint32_t a = perm[k], b = perm[k - 1];
        first = head[a] != head[b] || key2[a] != key2[b];
    }
    start[k] = first ? list[k] : 0ULL;
}

__global__ __launch_bounds__(SA_BLOCK) void k_big_write(const uint32_t *__restrict__ perm, const u64 *__restrict__ sa, const u64 *__restrict__ newhead,
                                                        const u64 *__restrict__ list, u64 m, u64 *__restrict__ SA, u64 *__restrict__ rank,
                                                        uint8_t *__restrict__ keep) {
    const u64 k = blockIdx.x * (u64)SA_BLOCK + threadIdx.x;
    if (k >= m) return;
    const u64 p = sa[perm[k]];
    SA[list[k]] = p;
    rank[p] = newhead[k];
    const bool head = newhead[k] == list[k];
    const bool next_head = k + 1 >= m || newhead[k + 1] == list[k + 1];
    keep[k] = (head && next_head) ? 0 : 1;                 // singletons leave the list
}

__global__ __launch_bounds__(SA_BLOCK) void k_big_bwt(const uint8_t *__restrict__ T, const u64 *__restrict__ SA, u64 first, u64 n, u64 nf, uint8_t *__restrict__ bw) {
    const u64 j = first + blockIdx.x * (u64)SA_BLOCK + threadIdx.x;
    if (j >= n) return;
    const u64 p = SA[j];
    bw[j] = (uint8_t)(T[p ? p - 1 : n - 1] | ((p >= nf && p < 2 * nf) ? 0x80 : 0));
}

namespace {
struct BigCtx { int device; };

int device_bwt_large(const uint8_t *T, uint64_t n, uint64_t nf, uint8_t *bw, void *ctx_) {
    const BigCtx *ctx = (const BigCtx *)ctx_;
    const bool verbose = getenv("NEWMAP_AMD_VERBOSE") && *getenv("NEWMAP_AMD_VERBOSE") != '0';
    if (n >= (1ULL << 40)) { nm_set_error("text too large for the device suffix sort"); return NM_E_TOO_LARGE; }
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreate(&st));
    auto fail = [&](int code) { (void)hipStreamDestroy(st); return code; };
#define TRYB(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { nm_set_error("HIP error %d (%s): %s", (int)e__, hipGetErrorString(e__), #expr); return fail(e__ == hipErrorOutOfMemory ? NM_E_ALLOC : NM_E_DEVICE); } } while (0)
    auto grid = [](u64 m) { return dim3((unsigned)((m + SA_BLOCK - 1) / SA_BLOCK)); };
    const u64 slice = 1ULL << 31;                            // positions per launch of the kernels that sweep the text
    DBuf dT, dSA, dRank, dHist, dCount;
    int rc;
    if ((rc = dev_alloc(dT, n)) || (rc = dev_alloc(dSA, n * 8)) || (rc = dev_alloc(dRank, n * 8)) || (rc = dev_alloc(dCount, 64))) {
        nm_set_error("not enough device memory for a text of %llu symbols", (unsigned long long)n);
        return fail(NM_E_ALLOC);
    }
    for (u64 a = 0; a < n; a += 1ULL << 30) {                 // copies of at most 1 GiB
        const u64 len = n - a < (1ULL << 30) ? n - a : (1ULL << 30);
        TRYB(hipMemcpyAsync((uint8_t *)dT.p + a, T + a, len, hipMemcpyHostToDevice, st));
    }
    const uint8_t *d_T = (const uint8_t *)dT.p;
    u64 *SA = (u64 *)dSA.p, *rank = (u64 *)dRank.p, *count = (u64 *)dCount.p;

    // ---- A. buckets of leading symbols: the fewest symbols whose largest bucket fits the working buffers
    size_t free_b = 0, total_b = 0;
    TRYB(hipMemGetInfo(&free_b, &total_b));
    const u64 budget = (u64)(free_b / 10 * 8) / 48;          // 48 bytes of buffers per member of a bucket
    const u64 hist_blocks = n / (16 * SA_BLOCK) + 1 < 16384 ? n / (16 * SA_BLOCK) + 1 : 16384;   // blocks of the text sweeps
    unsigned bits = 0;
    std::vector<u64> hist;
    for (unsigned try_bits : {6u, 9u, SA_MAX_BUCKET_BITS}) {
        const u64 nb = 1ULL << try_bits;
        DBuf dH;
        if ((rc = dev_alloc(dH, nb * 8)) != NM_OK) return fail(rc);
        TRYB(hipMemsetAsync(dH.p, 0, nb * 8, st));
        hipLaunchKernelGGL(k_big_hist, dim3((unsigned)hist_blocks), dim3(SA_BLOCK), 0, st, d_T, (u64)n, try_bits, (u64 *)dH.p);
        hist.assign(nb, 0);
        TRYB(hipMemcpyAsync(hist.data(), dH.p, nb * 8, hipMemcpyDeviceToHost, st));
        TRYB(hipStreamSynchronize(st));
        u64 mx = 0;
        for (u64 v : hist) mx = v > mx ? v : mx;
        if (verbose) fprintf(stderr, "[device sa] %u leading symbols: largest bucket %llu of %llu (room for %llu)\n", try_bits / 3, mx, (u64)n, budget);
        if (mx <= budget && mx < (1ULL << 31)) { bits = try_bits; break; }
    }
    if (!bits) { nm_set_error("device suffix sort: the text is too repetitive for bucketed sorting in the free device memory"); return fail(NM_E_TOO_LARGE); }
    u64 max_bucket = 0;
    for (u64 v : hist) max_bucket = v > max_bucket ? v : max_bucket;
    {
        DBuf dK0, dK1, dI0, dI1, dStart, dTmp;
        if ((rc = dev_alloc(dK0, max_bucket * 8)) || (rc = dev_alloc(dK1, max_bucket * 8)) || (rc = dev_alloc(dI0, max_bucket * 8)) ||
            (rc = dev_alloc(dI1, max_bucket * 8)) || (rc = dev_alloc(dStart, max_bucket * 8))) return fail(NM_E_ALLOC);
        rocprim::double_buffer<u64> keys((u64 *)dK0.p, (u64 *)dK1.p), vals((u64 *)dI0.p, (u64 *)dI1.p);
        size_t t_sort = 0, t_scan = 0;
        TRYB(rocprim::radix_sort_pairs(nullptr, t_sort, keys, vals, max_bucket, 0, 63, st));
        TRYB(rocprim::inclusive_scan(nullptr, t_scan, (u64 *)dStart.p, (u64 *)dStart.p, max_bucket, rocprim::maximum<u64>(), st));
        const size_t t_bytes = t_sort > t_scan ? t_sort : t_scan;
        if ((rc = dev_alloc(dTmp, t_bytes)) != NM_OK) return fail(NM_E_ALLOC);
        u64 offset = 0;
        for (u64 b = 0; b < hist.size(); b++) {
            const u64 m = hist[b];
            if (!m) continue;
            rocprim::double_buffer<u64> kb((u64 *)dK0.p, (u64 *)dK1.p), vb((u64 *)dI0.p, (u64 *)dI1.p);
            TRYB(hipMemsetAsync(count, 0, 8, st));
            hipLaunchKernelGGL(k_big_collect, dim3((unsigned)hist_blocks), dim3(SA_BLOCK), 0, st, d_T, (u64)n, bits, b,
                               vb.current(), kb.current(), count);
            if (verbose) {
                u64 got = 0;
                TRYB(hipMemcpyAsync(&got, count, 8, hipMemcpyDeviceToHost, st));
                TRYB(hipStreamSynchronize(st));
                if (got != m) { nm_set_error("device suffix sort: bucket %llu collected %llu of %llu", b, got, m); return fail(NM_E_DEVICE); }
            }
            size_t tb = t_bytes;
            TRYB(rocprim::radix_sort_pairs(dTmp.p, tb, kb, vb, m, 0, 63, st));
            hipLaunchKernelGGL(k_big_starts, grid(m), dim3(SA_BLOCK), 0, st, (const u64 *)kb.current(), m, offset, (u64 *)dStart.p);
            tb = t_bytes;
            TRYB(rocprim::inclusive_scan(dTmp.p, tb, (u64 *)dStart.p, (u64 *)dStart.p, m, rocprim::maximum<u64>(), st));
            hipLaunchKernelGGL(k_big_place, grid(m), dim3(SA_BLOCK), 0, st, (const u64 *)vb.current(), (const u64 *)dStart.p, m, offset, SA, rank);
            offset += m;
        }
        TRYB(hipStreamSynchronize(st));
        TRYB(hipGetLastError());
        if (offset != n) { nm_set_error("device suffix sort: bucket sizes do not add up"); return fail(NM_E_DEVICE); }
    }

    // ---- B. the tied groups
    u64 m = 0;
    DBuf dList;
    {
        DBuf dFlag, dTmp;
        if ((rc = dev_alloc(dFlag, n)) != NM_OK) return fail(NM_E_ALLOC);
        for (u64 first = 0; first < n; first += slice)
            hipLaunchKernelGGL(k_big_tied, grid(n - first < slice ? n - first : slice), dim3(SA_BLOCK), 0, st, (const u64 *)SA, (const u64 *)rank,
                               first, (u64)n, (uint8_t *)dFlag.p);
        // count first, so that the list gets exactly the room it needs; the primitives run over chunks of 2^30
        const u64 chunk = 1ULL << 30;
        size_t t_red = 0, t_sel = 0;
        TRYB(rocprim::reduce(nullptr, t_red, (uint8_t *)dFlag.p, count, 0ULL, chunk, rocprim::plus<u64>(), st));
        TRYB(rocprim::select(nullptr, t_sel, rocprim::counting_iterator<u64>(0), (uint8_t *)dFlag.p, (u64 *)nullptr, count, chunk, st));
        if ((rc = dev_alloc(dTmp, t_red > t_sel ? t_red : t_sel)) != NM_OK) return fail(NM_E_ALLOC);
        for (u64 a = 0; a < n; a += chunk) {
            const u64 len = n - a < chunk ? n - a : chunk;
            u64 part = 0;
            size_t tb = t_red;
            TRYB(rocprim::reduce(dTmp.p, tb, (uint8_t *)dFlag.p + a, count, 0ULL, len, rocprim::plus<u64>(), st));
            TRYB(hipMemcpyAsync(&part, count, 8, hipMemcpyDeviceToHost, st));
            TRYB(hipStreamSynchronize(st));
            m += part;
        }
        if (verbose) fprintf(stderr, "[device sa] h=%u: %llu of %llu suffixes still tied\n", SA_H0, m, (u64)n);
        if (m >= (1ULL << 31)) { nm_set_error("device suffix sort: %llu tied suffixes after %u symbols", m, SA_H0); return fail(NM_E_TOO_LARGE); }
        if (m) {
            if ((rc = dev_alloc(dList, m * 8)) != NM_OK) return fail(NM_E_ALLOC);
            u64 at = 0;
            for (u64 a = 0; a < n; a += chunk) {
                const u64 len = n - a < chunk ? n - a : chunk;
                u64 part = 0;
                size_t tb = t_sel;
                TRYB(rocprim::select(dTmp.p, tb, rocprim::counting_iterator<u64>(a), (uint8_t *)dFlag.p + a, (u64 *)dList.p + at, count, len, st));
                TRYB(hipMemcpyAsync(&part, count, 8, hipMemcpyDeviceToHost, st));
                TRYB(hipStreamSynchronize(st));
                at += part;
            }
            if (at != m) { nm_set_error("device suffix sort: tied-list size mismatch"); return fail(NM_E_DEVICE); }
        }
    }
    int rounds = 0;
    for (u64 h = SA_H0; m; h *= 2) {
        if (++rounds > 48 || h > 2 * (u64)n) { nm_set_error("device suffix sort did not converge"); return fail(NM_E_DEVICE); }
        DBuf dSa, dHead, dKey2, dKa, dKb, dPa, dPb, dStart, dKeep, dNew, dTmp;
        if ((rc = dev_alloc(dSa, m * 8)) || (rc = dev_alloc(dHead, m * 8)) || (rc = dev_alloc(dKey2, m * 8)) || (rc = dev_alloc(dKa, m * 8)) ||
            (rc = dev_alloc(dKb, m * 8)) || (rc = dev_alloc(dPa, m * 4)) || (rc = dev_alloc(dPb, m * 4)) || (rc = dev_alloc(dStart, m * 8)) ||
            (rc = dev_alloc(dKeep, m)) || (rc = dev_alloc(dNew, m * 8))) return fail(NM_E_ALLOC);
        u64 *list = (u64 *)dList.p;
        hipLaunchKernelGGL(k_big_gather, grid(m), dim3(SA_BLOCK), 0, st, (const u64 *)list, m, (const u64 *)SA, (const u64 *)rank, (u64)n, h,
                           (u64 *)dSa.p, (u64 *)dHead.p, (u64 *)dKey2.p, (uint32_t *)dPa.p);
        // stable sort by key2, then stable sort by group: members of a group end up ordered by key2, groups stay put
        rocprim::double_buffer<u64> kk((u64 *)dKa.p, (u64 *)dKb.p);
        rocprim::double_buffer<uint32_t> pp((uint32_t *)dPa.p, (uint32_t *)dPb.p);
        TRYB(hipMemcpyAsync(kk.current(), dKey2.p, m * 8, hipMemcpyDeviceToDevice, st));
        size_t t_sort = 0, t_scan = 0, t_sel = 0;
        TRYB(rocprim::radix_sort_pairs(nullptr, t_sort, kk, pp, m, 0, 64, st));
        TRYB(rocprim::inclusive_scan(nullptr, t_scan, (u64 *)dStart.p, (u64 *)dStart.p, m, rocprim::maximum<u64>(), st));
        TRYB(rocprim::select(nullptr, t_sel, list, (uint8_t *)dKeep.p, (u64 *)dNew.p, count, m, st));
        size_t t_bytes = t_sort > t_scan ? t_sort : t_scan;
        if (t_sel > t_bytes) t_bytes = t_sel;
        if ((rc = dev_alloc(dTmp, t_bytes)) != NM_OK) return fail(NM_E_ALLOC);
        unsigned nbits = 1;
        while (nbits < 64 && (n >> nbits)) nbits++;
        size_t tb = t_bytes;
        TRYB(rocprim::radix_sort_pairs(dTmp.p, tb, kk, pp, m, 0, nbits, st));
        hipLaunchKernelGGL(k_big_pick, grid(m), dim3(SA_BLOCK), 0, st, (const uint32_t *)pp.current(), (const u64 *)dHead.p, m, kk.current());
        tb = t_bytes;
        TRYB(rocprim::radix_sort_pairs(dTmp.p, tb, kk, pp, m, 0, nbits, st));
        hipLaunchKernelGGL(k_big_regroup, grid(m), dim3(SA_BLOCK), 0, st, (const uint32_t *)pp.current(), (const u64 *)dHead.p, (const u64 *)dKey2.p,
                           (const u64 *)list, m, (u64 *)dStart.p);
        tb = t_bytes;
        TRYB(rocprim::inclusive_scan(dTmp.p, tb, (u64 *)dStart.p, (u64 *)dStart.p, m, rocprim::maximum<u64>(), st));
        hipLaunchKernelGGL(k_big_write, grid(m), dim3(SA_BLOCK), 0, st, (const uint32_t *)pp.current(), (const u64 *)dSa.p, (const u64 *)dStart.p,
                           (const u64 *)list, m, SA, rank, (uint8_t *)dKeep.p);
        tb = t_bytes;
        TRYB(rocprim::select(dTmp.p, tb, list, (uint8_t *)dKeep.p, (u64 *)dNew.p, count, m, st));
        u64 left = 0;
        TRYB(hipMemcpyAsync(&left, count, 8, hipMemcpyDeviceToHost, st));
        TRYB(hipStreamSynchronize(st));
        TRYB(hipGetLastError());
        if (verbose) fprintf(stderr, "[device sa] h=%llu: %llu of %llu tied suffixes left\n", h * 2, left, m);
        if (left) TRYB(hipMemcpyAsync(list, dNew.p, left * 8, hipMemcpyDeviceToDevice, st));
        TRYB(hipStreamSynchronize(st));
        m = left;
    }

    // ---- C. the BWT column
    dRank.reset();
    DBuf dBw;
    if ((rc = dev_alloc(dBw, n)) != NM_OK) return fail(NM_E_ALLOC);
    for (u64 first = 0; first < n; first += slice)
        hipLaunchKernelGGL(k_big_bwt, grid(n - first < slice ? n - first : slice), dim3(SA_BLOCK), 0, st, d_T, (const u64 *)SA, first, (u64)n, (u64)nf,
                           (uint8_t *)dBw.p);
    for (u64 a = 0; a < n; a += 1ULL << 30) {
        const u64 len = n - a < (1ULL << 30) ? n - a : (1ULL << 30);
        TRYB(hipMemcpyAsync(bw + a, (const uint8_t *)dBw.p + a, len, hipMemcpyDeviceToHost, st));
    }
    TRYB(hipStreamSynchronize(st));
    if (uint8_t *lcp_host = nm_build_lcp_buffer()) {        // the LCP bytes while text and suffix array are here
        dBw.reset();
        DBuf dLcp;
        if ((rc = dev_alloc(dLcp, n + 1)) != NM_OK) return fail(NM_E_ALLOC);
        for (u64 first = 0; first <= n; first += slice)
            hipLaunchKernelGGL((k_lcp_bytes<u64>), grid(n + 1 - first < slice ? n + 1 - first : slice), dim3(SA_BLOCK), 0, st, d_T, (const u64 *)SA, first, (u64)n,
                               (uint8_t *)dLcp.p);
        for (u64 a = 0; a < n + 1; a += 1ULL << 30) {
            const u64 len = n + 1 - a < (1ULL << 30) ? n + 1 - a : (1ULL << 30);
            TRYB(hipMemcpyAsync(lcp_host + a, (const uint8_t *)dLcp.p + a, len, hipMemcpyDeviceToHost, st));
        }
        TRYB(hipStreamSynchronize(st));
        nm_build_lcp_done();
    }
    TRYB(hipGetLastError());
    (void)hipStreamDestroy(st);
    return NM_OK;
#undef TRYB
}
}  // namespace

extern "C" int nm_index_build_device(const char *fasta_path, const char *index_path, uint8_t sa_ratio, uint8_t seed_len, int device) {
    if (device < 0) { nm_set_error("device %d: the device builder needs a GPU (nm_index_build is the host builder)", device); return NM_E_DEVICE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) { nm_set_error("device %d requested but %d HIP device(s) are visible", device, ndev); return NM_E_DEVICE; }
    Ctx ctx{device};                                       // (BigCtx has the same layout)
    return nm_index_build_impl(fasta_path, index_path, sa_ratio, seed_len, device_suffix_array, &ctx, device_bwt_large);
}

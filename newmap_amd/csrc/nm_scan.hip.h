// nm_scan.hip.h -- hierarchical exclusive scan of uint64 arrays on the device (internal linkage, so
// that several translation units of libnewmap_amd.so may include it).
#ifndef NM_SCAN_HIP_H
#define NM_SCAN_HIP_H

#include <hip/hip_runtime.h>
#include <cstdint>

#include "../../include/newmap_amd.h"
#include "nm_internal.h"

#define TB 256                     // threads per block
#define TI 16                      // items per thread
#define TILE (TB * TI)             // items per block

// ---------------------------------------------------------------- exclusive scan of u64 -----
// level kernel: in-place exclusive scan of each TILE-sized tile, tile totals to sums[]
static __global__ __launch_bounds__(TB) void k_scan_tiles(uint64_t *__restrict__ data, uint64_t n, uint64_t *__restrict__ sums) {
    __shared__ uint64_t warp_tot[TB / 64];
    const uint64_t base = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * TI;
    uint64_t v[TI], run = 0;
#pragma unroll
    for (int j = 0; j < TI; j++) {
        const uint64_t x = base + j < n ? data[base + j] : 0;
        v[j] = run;                                   // exclusive within the thread
        run += x;
    }
    // exclusive scan of the per-thread totals across the block
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint64_t inc = run;
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t y = __shfl_up(inc, off, 64);
        if ((int)lane >= off) inc += y;
    }
    if (lane == 63) warp_tot[wv] = inc;
    __syncthreads();
    uint64_t before = inc - run;
    for (uint32_t w = 0; w < wv; w++) before += warp_tot[w];
#pragma unroll
    for (int j = 0; j < TI; j++)
        if (base + j < n) data[base + j] = v[j] + before;
    if (threadIdx.x == TB - 1 && sums) sums[blockIdx.x] = before + run;
}

static __global__ __launch_bounds__(TB) void k_scan_add(uint64_t *__restrict__ data, uint64_t n, const uint64_t *__restrict__ sums) {
    const uint64_t add = sums[blockIdx.x];
    const uint64_t base = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * TI;
#pragma unroll
    for (int j = 0; j < TI; j++)
        if (base + j < n) data[base + j] += add;
}

// exclusive scan of data[0..n) in place; *total (device) receives the sum.  scratch: >= n/TILE + 4096 u64
static int scan_exclusive(uint64_t *d_data, uint64_t n, uint64_t *d_scratch, uint64_t *d_total, hipStream_t st) {
    const uint64_t tiles = (n + TILE - 1) / TILE;
    if (tiles <= 1) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(TB), 0, st, d_data, n, d_total);
        HIP_TRY(hipGetLastError());
        return NM_OK;
    }
    hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)tiles), dim3(TB), 0, st, d_data, n, d_scratch);
    HIP_TRY(hipGetLastError());
    int rc = scan_exclusive(d_scratch, tiles, d_scratch + tiles, d_total, st);
    if (rc != NM_OK) return rc;
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(TB), 0, st, d_data, n, d_scratch);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}


#endif

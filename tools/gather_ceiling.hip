// tools/gather_ceiling.hip -- measurement tool (not part of the product): the rate at which one MI355X serves RANDOM
// small reads from a table far larger than its caches, the access pattern of k_sites (one 128-byte quad-table entry per
// group of positions, two 16-byte loads of it per lane, two entries in flight per lane).
//
//   gather_ceiling TABLE_GiB GRAN LOADS POLICY [LINES_LOG2] [WAVES_PER_CU] [SECOND_OFF] [PAGE_LOG2]
//     GRAN    bytes between the slots a request may start at: 128 (whole line, as the quad table), 64, 32
//     LOADS   16-byte loads per slot: 1; 2 = two load INSTRUCTIONS of one lane (the second SECOND_OFF bytes further: default
//             64 for GRAN 128, else 16); 3 = two LANES of one load instruction (lanes 2 j and 2 j + 1 share the slot and
//             read its bytes 0..15 and SECOND_OFF .. +15: the coalescer sees one line per lane pair)
//     POLICY  0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt   (gfx942/gfx950 cache-policy bits of global_load)
//     PAGE_LOG2  0 = every lane anywhere in the table; p > 0 = the 64 lanes of a wave take their slots of one round from ONE
//             aligned window of 2^p bytes (picked at random per wave and round): the most that bucketing a block's keys
//             by page could buy, without the cost of the bucketing
//
// Prints one JSON line: slots/s, bytes requested/s, lines/s x 128 B.  Run it under `rocprofv3 --pmc TCC_EA0_RDREQ_sum
// TCC_EA0_RDREQ_32B_sum ...` to see which request sizes the L2 sends to the fabric for each form.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(r_), __FILE__, __LINE__); exit(2); } } while (0)

template <int POLICY>
__device__ __forceinline__ v4u load16(const uint8_t *p) {
    v4u r;
    if (POLICY == 0) { r = *reinterpret_cast<const v4u *>(p); return r; }
    if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory");
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(r) : "v"(p) : "memory");
    return r;
}

__device__ __forceinline__ uint64_t mix(uint64_t z) {            // splitmix64
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// every lane: `iters` rounds of 2 slots in flight (as k_sites), LOADS 16-byte loads per slot
template <int POLICY, int LOADS>
__global__ __launch_bounds__(256) void k_gather(const uint8_t *__restrict__ table, uint64_t n_slots, uint32_t gran, uint32_t second_off,
                                                uint32_t iters, uint64_t *__restrict__ sink, uint64_t slots_per_page) {
    const uint64_t lane_id = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t gid = LOADS == 3 ? lane_id >> 1 : lane_id;      // LOADS == 3: a lane pair walks one sequence of slots
    uint64_t state = mix(gid * 0x2545F4914F6CDD1DULL + 1);
    uint64_t wstate = mix((lane_id >> 6) * 0x9E3779B97F4A7C15ULL + 7);      // the same in every lane of a wave
    const uint64_t n_pages = slots_per_page ? n_slots / slots_per_page : 0;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        v4u r[2][2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            state = mix(state);
            uint64_t slot = __umul64hi(state, n_slots);
            if (slots_per_page) { wstate = mix(wstate); slot = __umul64hi(wstate, n_pages) * slots_per_page + __umul64hi(state, slots_per_page); }
            const uint8_t *p = table + slot * gran + (LOADS == 3 && (lane_id & 1) ? second_off : 0u);
            r[s][0] = load16<POLICY>(p);
            r[s][1] = LOADS == 2 ? load16<POLICY>(p + second_off) : r[s][0];
        }
        if (POLICY != 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[1][0]), "+v"(r[1][1]) :: "memory");
#pragma unroll
        for (int s = 0; s < 2; s++) acc ^= r[s][0].x ^ r[s][0].w ^ r[s][1].y ^ r[s][1].z;
        // the next addresses depend on the data: no hoisting across rounds (a lane pair must stay in step: both take the pair's bit)
        state ^= (LOADS == 3 ? (uint32_t)__shfl(acc, (int)((threadIdx.x & 63) & ~1u), 64) : acc) & 1u;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;
}

__global__ void k_fill(uint64_t *p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = mix(i) & 0xFEFEFEFEFEFEFEFEULL;
}

template <int POLICY>
static void launch(int loads, dim3 g, dim3 b, const uint8_t *t, uint64_t n_slots, uint32_t gran, uint32_t off, uint32_t iters, uint64_t *sink, uint64_t spp) {
    if (loads == 3)      hipLaunchKernelGGL((k_gather<POLICY, 3>), g, b, 0, 0, t, n_slots, gran, off, iters, sink, spp);
    else if (loads == 2) hipLaunchKernelGGL((k_gather<POLICY, 2>), g, b, 0, 0, t, n_slots, gran, off, iters, sink, spp);
    else                 hipLaunchKernelGGL((k_gather<POLICY, 1>), g, b, 0, 0, t, n_slots, gran, off, iters, sink, spp);
}

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s TABLE_GiB GRAN LOADS POLICY [LINES_LOG2=28] [WAVES_PER_CU=32] [SECOND_OFF] [PAGE_LOG2=0]\n", argv[0]); return 1; }
    const double gib = atof(argv[1]);
    const uint32_t gran = (uint32_t)atoi(argv[2]);
    const int loads = atoi(argv[3]), policy = atoi(argv[4]);
    const int lines_log2 = argc > 5 ? atoi(argv[5]) : 28;
    const int waves_per_cu = argc > 6 ? atoi(argv[6]) : 32;
    if ((gran != 128 && gran != 64 && gran != 32) || loads < 1 || loads > 3 || policy < 0 || policy > 4) { fprintf(stderr, "bad arguments\n"); return 1; }
    const uint64_t bytes = (uint64_t)(gib * (double)(1ULL << 30)) / 4096 * 4096;
    uint8_t *table = nullptr;
    uint64_t *sink = nullptr;
    CHECK(hipMalloc(&table, bytes + 256));
    CHECK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)table, bytes / 8);
    CHECK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint64_t n_slots = bytes / gran;
    const uint32_t second_off = argc > 7 ? (uint32_t)atoi(argv[7]) : (gran == 128 ? 64 : 16);
    const int page_log2 = argc > 8 ? atoi(argv[8]) : 0;
    const uint64_t spp = page_log2 ? (1ULL << page_log2) / gran : 0;
    if (spp && spp > n_slots) { fprintf(stderr, "page larger than the table\n"); return 1; }
    const uint64_t lanes = (uint64_t)prop.multiProcessorCount * waves_per_cu * 64;
    const uint64_t total = 1ULL << lines_log2;
    const uint64_t walkers = loads == 3 ? lanes / 2 : lanes;      // sequences of slots
    const uint32_t iters = (uint32_t)(total / (walkers * 2)) ? (uint32_t)(total / (walkers * 2)) : 1;
    const dim3 grid((unsigned)(lanes / 256)), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {                           // rep 0 warms the TLBs / page tables
        CHECK(hipEventRecord(e0, 0));
        switch (policy) {
            case 0: launch<0>(loads, grid, block, table, n_slots, gran, second_off, iters, sink, spp); break;
            case 1: launch<1>(loads, grid, block, table, n_slots, gran, second_off, iters, sink, spp); break;
            case 2: launch<2>(loads, grid, block, table, n_slots, gran, second_off, iters, sink, spp); break;
            case 3: launch<3>(loads, grid, block, table, n_slots, gran, second_off, iters, sink, spp); break;
            default: launch<4>(loads, grid, block, table, n_slots, gran, second_off, iters, sink, spp); break;
        }
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double slots = (double)walkers * 2.0 * iters;
    static const char *pol[] = {"default", "nt", "sc1", "sc0 sc1", "sc1 nt"};
    printf("{\"table_gib\": %.2f, \"gran\": %u, \"loads_per_slot\": %d, \"second_off\": %u, \"policy\": \"%s\", \"page_log2\": %d, \"cus\": %d, \"waves_per_cu\": %d, \"slots\": %.0f, "
           "\"ms\": %.4f, \"g_slots_per_s\": %.2f, \"requested_gb_per_s\": %.1f, \"gb_per_s_at_128B_per_slot\": %.1f}\n",
           gib, gran, loads, second_off, pol[policy], page_log2, prop.multiProcessorCount, waves_per_cu, slots, best, slots / best / 1e6,
           slots * 16.0 * (loads == 3 ? 2 : loads) / best / 1e6, slots * 128.0 / best / 1e6);
    CHECK(hipFree(table));
    CHECK(hipFree(sink));
    return 0;
}

// nm_engine.hip -- the MI355X (gfx950 / CDNA4) engine behind the C-ABI of include/newmap_amd.h.
//
// Kernels (all integer / bit work, HBM-gather bound, no MFMA):
//   k_encode16 / k_encode   sequence bytes -> 2 bit-planes + ambiguity plane, 32 B / 64 bases
//   k_repeat_probe          one walk per 64 positions: settles long repeats, fixes lengths between equal ends
//   k_min_unique_quad       range mode, four positions per 32-byte quad-table entry (default, newmap/search.py:383-548)
//   k_min_unique_pair       range mode, two positions per pair-table block
//   k_min_unique            range mode, one lane per genome position (also --norc and short kmin)
//   k_min_unique_v2, _mp    earlier schedules of the same arithmetic, kept for A/B
//   k_fixed_k               list mode,  one lane per genome position (newmap/search.py:551-644)
//   k_multi                 several FASTA files x several index files (newmap/search.py:461, 656-697)
//   k_count                 forward-strand counts of (start, len) k-mers (src/newmap-count.c:91-206)
//   k_upper                 per-position upper search length (newmap/search.py:744-882)
//   k_seed, k_seed_level, k_quad_build, k_pair_gather, k_pair, k_lf_blocks, k_rank2_*   tables built at open
// The per-position logic lives in nm_core.h.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/newmap_amd.h"
#include "nm_format.h"
#include "nm_internal.h"

#define NM_HD __device__ __forceinline__
struct nm_view;
static __device__ __forceinline__ uint64_t nm_seed_load_policy(const nm_view &ix, uint64_t slot);
#define NM_SEED_LOAD(ix, slot) nm_seed_load_policy((ix), (slot))
#include "nm_core.h"

// seed-table gather under a selectable cache policy (NM_OPT_SEED_POLICY; measurement knob)
static __device__ __forceinline__ uint64_t nm_seed_load_policy(const nm_view &ix, uint64_t slot) {
    const uint64_t *p = ix.seed + slot;
    if (ix.seed_policy == 1) return __builtin_nontemporal_load(p);
    if (ix.seed_policy == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

#define NM_WAVE 64
#define NM_BLOCK 256

// ------------------------------------------------------------------------------ kernels ----

#define NM_WORK_WORDS 8             /* handle-owned counters: [0] chunk counter of k_min_unique_v2, [1..4] probe tally */

// the status words of a launch (and the handle's counters) start from zero; folded into the encode
// pass so that a segment costs one launch less (k_reset_status does the same on its own)
__device__ __forceinline__ void nm_reset_words(uint64_t *__restrict__ status, unsigned long long *__restrict__ work) {
    if (blockIdx.x == 0 && threadIdx.x < NM_STATUS_WORDS && status) status[threadIdx.x] = threadIdx.x == 2 ? ~0ULL : 0ULL;
    if (blockIdx.x == 0 && threadIdx.x < NM_WORK_WORDS && work) work[threadIdx.x] = 0ULL;
}

__global__ __launch_bounds__(NM_BLOCK) void k_encode(const uint8_t *__restrict__ seq, uint64_t seq_len,
                                                     nm_enc_word *__restrict__ enc, uint64_t n_words,
                                                     uint64_t *__restrict__ status, unsigned long long *__restrict__ work) {
    nm_reset_words(status, work);
    // one wave per 64-base word: three ballots give the three planes
    const uint64_t wave = (blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (wave >= n_words) return;
    const uint64_t pos = wave * 64 + lane;
    uint32_t code = 4;
    if (pos < seq_len) code = nm_base_code(seq[pos]);
    const uint64_t lo = __ballot((code & 1u) && code < 4);
    const uint64_t hi = __ballot((code & 2u) && code < 4);
    const uint64_t amb = __ballot(code > 3);
    if (lane == 0) {
        nm_enc_word w;
        w.lo = lo; w.hi = hi; w.amb = amb; w.pad = 0;
        enc[wave] = w;
    }
}

// 16 bases per lane (one 16-byte load), four lanes OR their 16-bit pieces into one 64-base word:
// 1 KiB per wave-instruction instead of the 64 B of k_encode.  Needs a 16-byte aligned `seq`.
__global__ __launch_bounds__(NM_BLOCK) void k_encode16(const uint8_t *__restrict__ seq, uint64_t seq_len,
                                                       nm_enc_word *__restrict__ enc, uint64_t n_words,
                                                       uint64_t *__restrict__ status, unsigned long long *__restrict__ work) {
    nm_reset_words(status, work);
    const uint64_t t = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (t >= n_words * 4) return;                       // groups of 4 lanes stay whole
    const uint64_t base = t * 16;
    uint32_t b[4] = {0, 0, 0, 0};
    uint32_t valid = 16;
    if (base + 16 <= seq_len) {
        const uint4 v = *reinterpret_cast<const uint4 *>(seq + base);
        b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
    } else {
        valid = base < seq_len ? (uint32_t)(seq_len - base) : 0;
        for (uint32_t j = 0; j < valid; j++) b[j >> 2] |= (uint32_t)seq[base + j] << (8 * (j & 3));
    }
    uint32_t lo = 0, hi = 0, amb = 0;                    // bytes past the end of the data are 0 in b[]: ambiguous
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        uint32_t l4, h4, a4;
        nm_base_codes4(b[j], l4, h4, a4);
        lo |= l4 << (4 * j);
        hi |= h4 << (4 * j);
        amb |= a4 << (4 * j);
    }
    const uint32_t sub = threadIdx.x & 3;
    uint64_t wlo = (uint64_t)lo << (16 * sub), whi = (uint64_t)hi << (16 * sub), wamb = (uint64_t)amb << (16 * sub);
    wlo |= __shfl_xor(wlo, 1, NM_WAVE);  whi |= __shfl_xor(whi, 1, NM_WAVE);  wamb |= __shfl_xor(wamb, 1, NM_WAVE);
    wlo |= __shfl_xor(wlo, 2, NM_WAVE);  whi |= __shfl_xor(whi, 2, NM_WAVE);  wamb |= __shfl_xor(wamb, 2, NM_WAVE);
    if (sub == 0) {
        nm_enc_word w;
        w.lo = wlo; w.hi = whi; w.amb = wamb; w.pad = 0;
        enc[t >> 2] = w;
    }
}

template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_seed(nm_view ix, uint64_t *__restrict__ table, uint64_t first_slot,
                                                   uint64_t n_slots, uint32_t s) {
    const uint64_t slot = first_slot + blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (slot < n_slots) table[slot] = nm_seed_entry<BIG>(ix, slot, s);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, NM_WAVE);
    return v;
}

// shared epilogue: ambiguous count, error report, optional tallies
template <bool STATS>
__device__ __forceinline__ void nm_epilogue(bool inb, bool amb0, bool err, uint64_t p, const nm_tally &t,
                                            uint64_t *__restrict__ status) {
    const uint64_t amb_mask = __ballot(inb && amb0);
    const uint64_t err_mask = __ballot(inb && err);
    const uint32_t lane = threadIdx.x & 63;
    if (amb_mask && lane == 0) atomicAdd((unsigned long long *)&status[0], (unsigned long long)__popcll(amb_mask));
    if (err_mask) {
        if (inb && err) atomicMin((unsigned long long *)&status[2], (unsigned long long)p);
        if (lane == 0) atomicOr((unsigned long long *)&status[1], 1ULL);
    }
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks), c = wave_sum(t.seeds),
                       d = wave_sum(t.strands);
        const uint32_t e = (uint32_t)__popcll(__ballot(inb && !amb0));
        if (lane == 0) {
            atomicAdd((unsigned long long *)&status[3], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)d);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)e);
        }
    }
}

__device__ __forceinline__ void nm_store(void *out, int elem_bytes, uint64_t p, uint32_t v) {
    if (elem_bytes == 1) ((uint8_t *)out)[p] = (uint8_t)v;
    else if (elem_bytes == 2) ((uint16_t *)out)[p] = (uint16_t)v;
    else ((uint32_t *)out)[p] = v;
}

// ---- k_repeat_probe: one lane per NM_PROBE_STRIDE positions (nm_core.h: nm_repeat_probe) -------
// Runs before the range kernel.  probe[j] = the word of nm_repeat_probe for stride j: how many positions from
// j * NM_PROBE_STRIDE on lie inside a stretch that occurs twice over at least kmax bases (the range kernel
// stores 0 for them) and the exact least unique length at the probe position (two neighbouring strides with the
// same end decide every position between them); such positions neither read a table line nor walk.  Lanes of
// a wave probe neighbouring strides, so inside a long repeat they walk in step.  probe[n_probes] = 0 (the
// consumers read strides j and j+1).  probe_tally (counter builds): LF steps, blocks, seed entries, settled.
#define NM_PROBE_STRIDE 64u
// coarse[c] = positions from c * NM_COARSE_STRIDE on that one walk of <= kmax + NM_COARSE_STRIDE - 1 bases settles as 0
template <bool BIG, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_repeat_probe_coarse(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t n_coarse,
                                                                  uint32_t kmax, uint32_t *__restrict__ coarse,
                                                                  unsigned long long *__restrict__ probe_tally) {
    const uint64_t c = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    nm_tally t = {0, 0, 0, 0};
    if (c < n_coarse) {
        uint32_t settled, exact;
        nm_repeat_probe_ex<BIG>(ix, enc, c * NM_COARSE_STRIDE, kmax, NM_COARSE_STRIDE, t, settled, exact);
        coarse[c] = settled;
    }
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks), d = wave_sum(t.seeds);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&probe_tally[0], (unsigned long long)a);
            atomicAdd(&probe_tally[1], (unsigned long long)b);
            atomicAdd(&probe_tally[2], (unsigned long long)d);
        }
    }
}

template <bool BIG, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_repeat_probe(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t n_probes,
                                                           uint32_t kmax, uint32_t *__restrict__ probe,
                                                           unsigned long long *__restrict__ probe_tally,
                                                           const uint32_t *__restrict__ coarse, volatile uint32_t *repeats_seen,
                                                           uint32_t *__restrict__ seen_latch) {
    const uint64_t j = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    nm_tally t = {0, 0, 0, 0};
    uint32_t c = 0;
    if (j <= n_probes) {
        uint32_t word = 0;
        if (j < n_probes) {
            const uint64_t P = j * NM_PROBE_STRIDE;
            // a stride the coarse probe settles completely: the word this probe would find after kmax + 63 steps
            if (coarse && nm_coarse_covers(coarse[P / NM_COARSE_STRIDE], (uint32_t)(P % NM_COARSE_STRIDE), NM_PROBE_STRIDE)) word = NM_PROBE_STRIDE;
            else word = nm_repeat_probe<BIG>(ix, enc, P, kmax, NM_PROBE_STRIDE, t);
        }
        probe[j] = word;
        c = word & 0xFFu;
    }
    // tell the host (a word of pinned, device-visible memory, read without synchronisation before later launches)
    // that this input has stretches repeated over more than kmax bases: the coarse probes then pay off
    // (once per handle: a latch in device memory keeps later blocks from writing across PCIe again)
    // (wave by wave -- a block barrier here would hold finished waves' slots until the longest walk of the block ends)
    if (repeats_seen && __ballot(c == NM_PROBE_STRIDE) && (threadIdx.x & 63) == 0 && *seen_latch == 0u) {
        *seen_latch = 1u;
        *repeats_seen = 1u;
    }
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks), d = wave_sum(t.seeds), e = wave_sum(c);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&probe_tally[0], (unsigned long long)a);
            atomicAdd(&probe_tally[1], (unsigned long long)b);
            atomicAdd(&probe_tally[2], (unsigned long long)d);
            atomicAdd(&probe_tally[3], (unsigned long long)e);
        }
    }
}

template <bool BIG, bool RC, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_min_unique(nm_view ix, const nm_enc_word *__restrict__ enc,
                                                         uint64_t num_kmers, uint32_t kmin, uint32_t kmax,
                                                         void *__restrict__ out, int elem_bytes,
                                                         uint64_t *__restrict__ status,
                                                         const uint32_t *__restrict__ probe) {
    const uint64_t p = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    const bool inb = p < num_kmers;
    bool amb0 = false, err = false;
    nm_tally t = {0, 0, 0, 0};
    uint32_t r = 0;
    if (inb) {
        // positions the repeat probes decide (k_repeat_probe) are stored without touching the index
        uint32_t ks = NM_PROBE_OPEN;
        if (probe) ks = nm_probe_kstar(probe[p / NM_PROBE_STRIDE], probe[p / NM_PROBE_STRIDE + 1], (uint32_t)(p & (NM_PROBE_STRIDE - 1)), NM_PROBE_STRIDE, kmax);
        if (ks == NM_PROBE_OPEN) {
            r = nm_min_unique_one<BIG, RC>(ix, enc, p, kmin, kmax, amb0, err, t);
        } else {
            nm_window w = nm_load_window(enc, p);
            uint32_t kbase = 0;
            r = nm_probe_element(ks, kmin, kmax, ks < kmin && nm_all_valid(enc, p, w, kbase, 0, kmin));
        }
        nm_store(out, elem_bytes, p, r);
    }
    nm_epilogue<STATS>(inb, amb0, err, p, t, status);
}


// ---- k_min_unique_mp: NM_MP positions per lane --------------------------------------------------
// Same arithmetic as k_min_unique.  Two changes in how the work is laid out on a wave:
//   * a wave owns 64*NM_MP consecutive positions, so the encoded words its windows are cut from are
//     wave-uniform: NM_MP+1 scalar 32-byte loads replace 4 vector loads per lane and position;
//   * a lane first issues the seed-table lookups of all its NM_MP positions (independent HBM
//     gathers in flight together) and only then consumes them, walking further where needed.
#define NM_MP 4

template <bool BIG, bool RC, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_min_unique_mp(nm_view ix, const nm_enc_word *__restrict__ enc,
                                                            uint64_t n_enc_words, uint64_t num_kmers,
                                                            uint32_t kmin, uint32_t kmax, void *__restrict__ out,
                                                            int elem_bytes, uint64_t *__restrict__ status) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t wave_base = ((uint64_t)blockIdx.x * (NM_BLOCK / NM_WAVE) + wave_in_block) * (64ull * NM_MP);
    if (wave_base >= num_kmers) return;                                  // whole wave out of range
    const uint32_t s = ix.seed_len;
    const bool use_seed = s && kmin >= s;

    // wave-uniform words (scalar loads); indexes beyond the array are clamped to the all-ambiguous pad
    nm_enc_word W[NM_MP + 1];
    const uint64_t w0 = wave_base >> 6;
#pragma unroll
    for (int j = 0; j <= NM_MP; j++) {
        uint64_t wi = w0 + j;
        if (wi >= n_enc_words) wi = n_enc_words - 1;
        W[j] = enc[wi];
    }
    nm_window win[NM_MP];
    uint64_t e[NM_MP];
    bool amb0[NM_MP], settled[NM_MP];
    uint32_t n_amb = 0, n_searched = 0;
    nm_tally t = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NM_MP; j++) {
        const uint64_t p = wave_base + 64ull * j + lane;
        win[j] = nm_window_from(W[j], W[j + 1], lane);
        settled[j] = nm_min_unique_settled(win[j], s, use_seed, amb0[j]);
        if (p >= num_kmers) { settled[j] = true; amb0[j] = false; }
        e[j] = 0;
        if (!settled[j] && use_seed) e[j] = NM_SEED_LOAD(ix, nm_seed_slot(win[j], s));   // NM_MP gathers in flight
    }
    bool any_err = false;
    uint64_t err_pos = ~0ULL;
#pragma unroll
    for (int j = 0; j < NM_MP; j++) {
        const uint64_t p = wave_base + 64ull * j + lane;
        if (p >= num_kmers) continue;
        uint32_t r = 0;
        if (amb0[j]) n_amb++;
        if (!settled[j]) {
            uint64_t lo = 0, hi = ix.n;
            uint32_t k = 0;
            if (use_seed) {
                if (STATS) t.seeds++;
                if (nm_seed_decode(e[j], lo, hi)) k = s;
                else { lo = 0; hi = ix.n; }
            }
            bool err = false;
            r = nm_min_unique_walk_any<BIG, RC>(ix, enc, p, win[j], lo, hi, k, kmin, kmax, err, t);
            if (err) { any_err = true; if (p < err_pos) err_pos = p; }
        }
        if (STATS && !amb0[j]) n_searched++;
        nm_store(out, elem_bytes, p, r);
    }
    // wave totals
    const uint32_t amb_sum = wave_sum(n_amb);
    if (lane == 0 && amb_sum) atomicAdd((unsigned long long *)&status[0], (unsigned long long)amb_sum);
    if (__ballot(any_err)) {
        if (any_err) atomicMin((unsigned long long *)&status[2], (unsigned long long)err_pos);
        if (lane == 0) atomicOr((unsigned long long *)&status[1], 1ULL);
    }
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks), c = wave_sum(t.seeds),
                       d = wave_sum(t.strands), f = wave_sum(n_searched);
        if (lane == 0) {
            atomicAdd((unsigned long long *)&status[3], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)d);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)f);
        }
    }
}

// ---- k_min_unique_pair: one 128-byte line serves TWO positions ---------------------------------
// Measured (profiles/round1): with a long seed table the range kernel is bound by HBM line fetches --
// every L2 miss is a 128-byte read (TCC_EA0_RDREQ_128B), one per position for its 8-byte seed entry,
// ~5.8 TB/s of real traffic whatever the kernel's structure.  The only lever left is fewer lines per
// position.  Neighbouring positions p and p+1 share the m-mer core Y = S[p+1 .. p+1+m): their
// (m+1)-mers are S[p].Y and Y.S[p+1+m].  The pair table stores, per core, the intervals of all four
// a.Y and all four Y.b in one 64-byte block, so a lane that owns positions (2i, 2i+1) touches ONE line
// for both seeds.
template <bool BIG, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_min_unique_pair(nm_view ix, const nm_enc_word *__restrict__ enc,
                                                              uint64_t n_enc_words, uint64_t num_kmers,
                                                              uint32_t kmin, uint32_t kmax, void *__restrict__ out,
                                                              int elem_bytes, uint64_t *__restrict__ status,
                                                              const uint32_t *__restrict__ probe) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t wave_base = ((uint64_t)blockIdx.x * (NM_BLOCK / NM_WAVE) + wave_in_block) * 128ull;
    // (no early exit for waves past the end: every wave of the block reaches the barriers below)
    const uint32_t m = ix.pair_m, s = m + 1;
    const uint64_t core_mask = (1ULL << m) - 1ULL;

    nm_enc_word W[3];                                     // wave-uniform -> scalar loads
    const uint64_t w0 = wave_base >> 6;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        uint64_t wi = w0 + j;
        if (wi >= n_enc_words) wi = n_enc_words - 1;
        W[j] = enc[wi];
    }
    const uint32_t q = 2 * lane;                          // offset of the even position in the wave's 128
    const nm_enc_word &Wa = q < 64 ? W[0] : W[1];
    const nm_enc_word &Wb = q < 64 ? W[1] : W[2];
    const uint64_t p0 = wave_base + q, p1 = p0 + 1;
    const nm_window win0 = nm_window_from(Wa, Wb, q & 63);
    const nm_window win1 = nm_window_from(Wa, Wb, (q & 63) + 1);   // (q & 63) <= 62
    const bool in0 = p0 < num_kmers, in1 = p1 < num_kmers;
    const bool amb0 = (win0.amb & 1ULL) != 0, amb1 = (win0.amb & 2ULL) != 0;
    const bool core_ok = ((win0.amb >> 1) & core_mask) == 0;
    // positions the repeat probes decide store their element and read nothing (p0, p1 share a probe stride)
    uint32_t ks0 = NM_PROBE_OPEN, ks1 = NM_PROBE_OPEN;
    if (probe && in0) {
        const uint32_t wj = probe[p0 / NM_PROBE_STRIDE], wj1 = probe[p0 / NM_PROBE_STRIDE + 1];
        const uint32_t off0 = (uint32_t)(p0 & (NM_PROBE_STRIDE - 1));
        ks0 = nm_probe_kstar(wj, wj1, off0, NM_PROBE_STRIDE, kmax);
        ks1 = nm_probe_kstar(wj, wj1, off0 + 1, NM_PROBE_STRIDE, kmax);
    }
    const bool go0 = in0 && !amb0 && core_ok && ks0 == NM_PROBE_OPEN;                        // bases 0..m unambiguous
    const bool go1 = in1 && core_ok && !((win0.amb >> s) & 1ULL) && ks1 == NM_PROBE_OPEN;     // bases 1..m+1 unambiguous
    const uint64_t slot = ((win0.lo >> 1) & core_mask) | (((win0.hi >> 1) & core_mask) << m);
    const uint64_t *blk = ix.pair + slot * 8;
    uint64_t e0 = 0, e1 = 0;
    if (go0) e0 = blk[nm_window_code(win0, 0)];                                  // both in one 64-byte block
    if (go1) e1 = blk[4 + nm_window_code(win0, s)];
    nm_tally t = {0, 0, 0, 0};
    bool any_err = false;
    uint64_t err_pos = ~0ULL;
    // ---- stage 1: what the table line alone decides.  Most positions end here (interval of one
    // element); the rest are queued in LDS so that only as many waves as there is work stay resident
    __shared__ uint64_t q_p[NM_BLOCK * 2], q_lo[NM_BLOCK * 2];
    __shared__ uint32_t q_cnt[NM_BLOCK * 2];
    __shared__ uint32_t q_n;
    if (threadIdx.x == 0) q_n = 0;
    __syncthreads();
    auto stage1 = [&](bool go, bool in_range, uint64_t p, const nm_window &win, uint64_t e, uint32_t ks) {
        if (!in_range) return;
        uint32_t r = 0;
        if (ks != NM_PROBE_OPEN) {                         // decided by the probes
            nm_window w = win;
            uint32_t kbase = 0;
            r = nm_probe_element(ks, kmin, kmax, ks < kmin && nm_all_valid(enc, p, w, kbase, 0, kmin));
        } else if (go) {
            uint64_t lo = 0, hi = ix.n;
            const bool have = nm_seed_decode(e, lo, hi);
            const uint64_t cnt = hi - lo;
            if (have && cnt == 1) {                       // unique already: max(s, kmin) if within U_p
                nm_window w = win;
                uint32_t kbase = 0;
                const uint32_t ans = s > kmin ? s : kmin;
                r = nm_all_valid(enc, p, w, kbase, s, ans) ? ans : 0u;
            } else if (have && cnt == 0) {                // search.py:699-722
                any_err = true;
                if (p < err_pos) err_pos = p;
            } else {                                      // needs a walk: queue it
                const uint32_t slot_i = atomicAdd(&q_n, 1u);
                q_p[slot_i] = p;
                q_lo[slot_i] = have ? lo : 0;
                q_cnt[slot_i] = have ? (uint32_t)cnt : 0u; // 0 = saturated entry, walk from scratch
                return;                                   // stored by stage 2
            }
        }
        nm_store(out, elem_bytes, p, r);
    };
    stage1(go0, in0, p0, win0, e0, ks0);
    stage1(go1, in1, p1, win1, e1, ks1);
    __syncthreads();
    // ---- stage 2: dense walks; waves beyond the queue length leave and free their slots
    const uint32_t n_walk = q_n;
    for (uint32_t i = threadIdx.x; i < n_walk; i += NM_BLOCK) {
        const uint64_t p = q_p[i];
        uint64_t lo = q_lo[i], hi = lo + q_cnt[i];
        uint32_t k = s;
        if (q_cnt[i] == 0) { lo = 0; hi = ix.n; k = 0; }
        bool err = false;
        const nm_window w = nm_load_window(enc, p);
        const uint32_t r = nm_min_unique_walk_any<BIG, true>(ix, enc, p, w, lo, hi, k, kmin, kmax, err, t);
        if (err) { any_err = true; if (p < err_pos) err_pos = p; }
        nm_store(out, elem_bytes, p, r);
    }

    const uint32_t amb_sum = wave_sum((uint32_t)(in0 && amb0) + (uint32_t)(in1 && amb1));
    if (lane == 0 && amb_sum) atomicAdd((unsigned long long *)&status[0], (unsigned long long)amb_sum);
    if (__ballot(any_err)) {
        if (any_err) atomicMin((unsigned long long *)&status[2], (unsigned long long)err_pos);
        if (lane == 0) atomicOr((unsigned long long *)&status[1], 1ULL);
    }
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks),
                       c = wave_sum((uint32_t)go0 + (uint32_t)go1),                 // 8-byte entries read
                       f = wave_sum((uint32_t)(in0 && !amb0) + (uint32_t)(in1 && !amb1));
        if (lane == 0) {
            atomicAdd((unsigned long long *)&status[3], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)f);
        }
    }
}


// ---- k_min_unique_quad: one 128-byte line serves FOUR positions ---------------------------------
// The pair kernel is bound by the table lines it fetches (one per two positions).  When kmin is at
// least w = m + 3, all a position needs from the table is ONE BIT -- "the w-mer here occurs once" --
// because the element stored is then kmin itself.  The quad table (nm_core.h) packs those bits for the
// four positions that share an m-mer core into one 32-byte entry: a lane owns positions 4i .. 4i+3,
// reads one entry, and only the positions whose w-mer is repeated (or absent) go on to the seed table
// and the walk (stage 2, compacted through LDS like the pair kernel's).
// A lane owns NM_QUAD_GROUPS groups of four positions (group g of lane l: wave base + 256 g + 4 l) and issues
// the entry loads of all its groups before it looks at any: with 32 waves per CU the requests in flight,
// not the lines per position, were what kept the kernel below the HBM's random-line rate.
#define NM_QUAD_GROUPS 2
#define NM_QUAD_MAX_KMIN 124u     /* kmin bases from any of a lane's four positions lie inside its two 64-base windows */
#define NM_QUAD_PER_WAVE (256u * NM_QUAD_GROUPS)
// Workgroup size (template argument QB): 256 lanes share one walk queue, so the walks of a stretch of repeated
// positions (the last kmax positions of every repeat) spread over four waves instead of running as four passes
// of one; 64 saves the barriers' waiting on unique input (2 % there, measured) -- NEWMAP_AMD_QUAD_BLOCK.
// LIST: list mode with several lengths, all of them >= m + 3 (newmap/search.py:551-644).  kmin = the FIRST listed
// length, kmax = the longest: a position whose (m+3)-mer occurs once is unique at every listed length, so its
// element is the first one (if its bases are unambiguous); positions the probes settle as repeated over more than
// kmax bases are 0; everything else -- repeated windows and lengths fixed by the probes -- goes through
// nm_fixed_k_one in stage 2.  The caller keeps the positions whose longest k-mer leaves the data out of this kernel.
template <bool BIG, bool STATS, int NM_QUAD_BLOCK, bool LONGK, bool LIST = false>
__global__ __launch_bounds__(NM_QUAD_BLOCK) void k_min_unique_quad(nm_view ix, const nm_enc_word *__restrict__ enc,
                                                              uint64_t n_enc_words, uint64_t num_kmers,
                                                              uint32_t kmin, uint32_t kmax, void *__restrict__ out,
                                                              int elem_bytes, uint64_t *__restrict__ status,
                                                              const uint32_t *__restrict__ probe,
                                                              uint64_t seq_len = 0, const uint32_t *__restrict__ list = nullptr,
                                                              uint32_t n_list = 0) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t wave_base = ((uint64_t)blockIdx.x * (NM_QUAD_BLOCK / NM_WAVE) + wave_in_block) * NM_QUAD_PER_WAVE;
    const uint32_t m = ix.quad_m;

    // the 4 * NM_QUAD_GROUPS + 1 encoded words of the wave: wave-uniform -> scalar loads.  Separate variables,
    // not an array: a lane picks its words by (lane >> 4), and selects over array elements are turned into an
    // indexed load from scratch memory
    const uint64_t w0 = wave_base >> 6;
#define NM_LOAD_WORD(j) const nm_enc_word W##j = enc[w0 + j < n_enc_words ? w0 + j : n_enc_words - 1];
    NM_LOAD_WORD(0) NM_LOAD_WORD(1) NM_LOAD_WORD(2) NM_LOAD_WORD(3) NM_LOAD_WORD(4)
    NM_LOAD_WORD(5) NM_LOAD_WORD(6) NM_LOAD_WORD(7) NM_LOAD_WORD(8) NM_LOAD_WORD(9)
#undef NM_LOAD_WORD
    static_assert(NM_QUAD_GROUPS == 2, "the word variables above are written out for two groups");
    const uint32_t q = 4 * lane;                          // offset of the lane's first position in a group's 256
    const uint32_t qw = q >> 6;
    // go: the position takes part in the lookup.  kmin <= NM_QUAD_MAX_KMIN (the launcher sees to it), so the
    // lane's 64-base window -- plus the ambiguity plane of the next 64 bases when kmin > 60 -- shows whether the
    // first kmin bases are free of ambiguity (if not: U_p < kmin, element 0, search.py:437); stage 1 is branch-free.
    constexpr bool long_kmin = LONGK;                      // kmin > 60: its own instantiation (6 more VGPRs cost a wave per SIMD)
    nm_window win[NM_QUAD_GROUPS];
    uint64_t amb_next[NM_QUAD_GROUPS];                     // ambiguity plane of bases p0+64 .. p0+127 (long_kmin only)
    auto kmin_bases_valid = [&](int g, uint32_t i) -> bool {
        const uint64_t a = win[g].amb >> i;
        if (!long_kmin) return (a & ((1ULL << kmin) - 1ULL)) == 0;
        const uint32_t n_lo = kmin < 64u - i ? kmin : 64u - i, n_hi = kmin - n_lo;        // n_hi <= 63
        const uint64_t lo_mask = n_lo == 64 ? ~0ULL : ((1ULL << n_lo) - 1ULL);
        return (a & lo_mask) == 0 && (amb_next[g] & ((1ULL << n_hi) - 1ULL)) == 0;
    };
#define NM_SEL4(f, a, b, c, d) (qw == 0 ? a.f : (qw == 1 ? b.f : (qw == 2 ? c.f : d.f)))
    uint64_t e[NM_QUAD_GROUPS][4];
    uint32_t go[NM_QUAD_GROUPS], inb[NM_QUAD_GROUPS];     // bit i: position i of the group
    uint32_t pw[NM_QUAD_GROUPS][2];                       // probe words of the group's stride and of the next one
    uint32_t n_amb = 0, n_searched = 0, n_entries = 0;
#pragma unroll
    for (int g = 0; g < NM_QUAD_GROUPS; g++) {
        {
            nm_enc_word Wa, Wb;
            Wa.pad = Wb.pad = 0;
            if (g == 0) {
                Wa.lo = NM_SEL4(lo, W0, W1, W2, W3); Wa.hi = NM_SEL4(hi, W0, W1, W2, W3); Wa.amb = NM_SEL4(amb, W0, W1, W2, W3);
                Wb.lo = NM_SEL4(lo, W1, W2, W3, W4); Wb.hi = NM_SEL4(hi, W1, W2, W3, W4); Wb.amb = NM_SEL4(amb, W1, W2, W3, W4);
            } else {
                Wa.lo = NM_SEL4(lo, W4, W5, W6, W7); Wa.hi = NM_SEL4(hi, W4, W5, W6, W7); Wa.amb = NM_SEL4(amb, W4, W5, W6, W7);
                Wb.lo = NM_SEL4(lo, W5, W6, W7, W8); Wb.hi = NM_SEL4(hi, W5, W6, W7, W8); Wb.amb = NM_SEL4(amb, W5, W6, W7, W8);
            }
            win[g] = nm_window_from(Wa, Wb, q & 63);      // bases p0 .. p0+63
            amb_next[g] = 0;
            if (long_kmin) {
                const uint64_t c = g == 0 ? NM_SEL4(amb, W2, W3, W4, W5) : NM_SEL4(amb, W6, W7, W8, W9);
                const uint32_t sh = q & 63;
                amb_next[g] = sh ? (Wb.amb >> sh) | (c << (64 - sh)) : Wb.amb;
            }
        }
        const uint64_t p0 = wave_base + 256u * g + q;
        uint32_t wj = 0, wj1 = 0;                          // four positions, one probe stride (words 0, 0: nothing decided)
        if (probe && p0 < num_kmers) { wj = probe[p0 / NM_PROBE_STRIDE]; wj1 = probe[p0 / NM_PROBE_STRIDE + 1]; }
        pw[g][0] = wj; pw[g][1] = wj1;
        const uint32_t off0 = (uint32_t)(p0 & (NM_PROBE_STRIDE - 1));
        go[g] = 0; inb[g] = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const bool in = p0 + i < num_kmers;
            const bool amb = ((win[g].amb >> i) & 1ULL) != 0;
            const bool kmin_valid = kmin_bases_valid(g, i);
            const uint32_t ks = nm_probe_kstar(wj, wj1, off0 + i, NM_PROBE_STRIDE, kmax);
            inb[g] |= (uint32_t)in << i;
            n_amb += (uint32_t)(in && amb);
            n_searched += (uint32_t)(in && !amb);
            go[g] |= (uint32_t)(in && kmin_valid && ks == NM_PROBE_OPEN) << i;
        }
        e[g][0] = e[g][1] = e[g][2] = e[g][3] = 0;
        if (go[g] && !(ix.seed_policy & 0x200u)) {        // the core lies inside every window that is free of ambiguity
            const ulonglong2 *ep = reinterpret_cast<const ulonglong2 *>(ix.quad + nm_quad_slot(win[g], m) * 4);
            const ulonglong2 a = ep[0], b = ep[1];
            e[g][0] = a.x; e[g][1] = a.y; e[g][2] = b.x; e[g][3] = b.y;
            n_entries += 4;
        }
    }
    nm_tally t = {0, 0, 0, 0};
    bool any_err = false;
    uint64_t err_pos = ~0ULL;
    __shared__ uint64_t q_p[NM_QUAD_BLOCK * 4 * NM_QUAD_GROUPS];
    __shared__ uint32_t q_n;
    if (threadIdx.x == 0) q_n = 0;
    __syncthreads();
    // ---- stage 1: what the entry alone decides.  once: least unique length <= w <= kmin, the element is kmin
    // (if within U_p); otherwise the w-mer is repeated or absent: seed table + walk in stage 2
#pragma unroll
    for (int g = 0; g < NM_QUAD_GROUPS; g++) {
        const uint64_t p0 = wave_base + 256u * g + q;
        const uint32_t once = nm_quad_bits(win[g], m, e[g]);
        uint32_t r[4] = {0, 0, 0, 0};
        const uint32_t hit = go[g] & once;
        uint32_t walk = go[g] & ~once;
        const uint32_t off0 = (uint32_t)(p0 & (NM_PROBE_STRIDE - 1));
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            r[i] = ((hit >> i) & 1u) ? kmin : 0u;
            const uint32_t ks = nm_probe_kstar(pw[g][0], pw[g][1], off0 + i, NM_PROBE_STRIDE, kmax);
            if (ks == NM_PROBE_OPEN) continue;
            if (!LIST) r[i] = nm_probe_element(ks, kmin, kmax, kmin_bases_valid(g, i));   // decided by the probes
            else if (ks <= kmax && ((inb[g] >> i) & 1u)) walk |= 1u << i;                  // (list mode: stage 2 picks the length)
        }
        if (walk) {
            uint32_t at = atomicAdd(&q_n, (uint32_t)__builtin_popcount(walk));
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if ((walk >> i) & 1u) q_p[at++] = p0 + i;
        }
        if (elem_bytes == 1 && inb[g] == 0xFu && (((uintptr_t)out) & 3u) == 0) {
            reinterpret_cast<uint32_t *>(out)[p0 >> 2] = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24);
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 4; i++)
                if ((inb[g] >> i) & 1u) nm_store(out, elem_bytes, p0 + i, r[i]);
        }
    }
    __syncthreads();                                       // stage 2 overwrites the placeholders of queued positions
    // ---- stage 2: dense walks
    const uint32_t n_walk = (ix.seed_policy & 0x100u) ? 0u : q_n;     // (0x100 / 0x200: timing experiments, wrong results)
    for (uint32_t i = threadIdx.x; i < n_walk; i += NM_QUAD_BLOCK) {
        const uint64_t p = q_p[i];
        bool amb0 = false, err = false;
        const uint32_t v = LIST ? nm_fixed_k_one<BIG, true>(ix, enc, p, seq_len, list, n_list, amb0, err, t)
                                : nm_min_unique_one<BIG, true>(ix, enc, p, kmin, kmax, amb0, err, t);
        if (err) { any_err = true; if (p < err_pos) err_pos = p; }
        nm_store(out, elem_bytes, p, v);
    }

    const uint32_t amb_sum = wave_sum(n_amb);
    if (lane == 0 && amb_sum) atomicAdd((unsigned long long *)&status[0], (unsigned long long)amb_sum);
    if (__ballot(any_err)) {
        if (any_err) atomicMin((unsigned long long *)&status[2], (unsigned long long)err_pos);
        if (lane == 0) atomicOr((unsigned long long *)&status[1], 1ULL);
    }
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks),
                       c = wave_sum(n_entries + t.seeds),                            // 8-byte table words read
                       f = wave_sum(n_searched);
        if (lane == 0) {
            atomicAdd((unsigned long long *)&status[3], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)f);
        }
    }
#undef NM_SEL4
}

// quad table from the seed table of the same length (nm_core.h: nm_quad_build_one)
template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_quad_build(nm_view ix, uint64_t *__restrict__ quad, uint64_t first_slot,
                                                         uint64_t n_slots, uint32_t m) {
    const uint64_t slot = first_slot + blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (slot < n_slots) nm_quad_build_one<BIG>(ix, slot, m, quad);
}

// level s of the seed table from level s-1 (one LF step per entry instead of s)
template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_seed_level(nm_view ix, const uint64_t *__restrict__ parent, uint64_t *__restrict__ table,
                                                         uint64_t first_slot, uint64_t n_slots, uint32_t s) {
    const uint64_t slot = first_slot + blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (slot < n_slots) table[slot] = nm_seed_entry_from_parent<BIG>(ix, parent[nm_seed_parent_slot(slot, s)], slot, s);
}

// the pair table is a rearrangement of the seed table of level m+1
__global__ __launch_bounds__(NM_BLOCK) void k_pair_gather(const uint64_t *__restrict__ seed, uint64_t *__restrict__ table,
                                                          uint64_t first, uint64_t n_entries, uint32_t m) {
    const uint64_t i = first + blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (i < n_entries) table[i] = seed[nm_pair_seed_slot(i >> 3, m, (uint32_t)(i & 7))];
}

template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_pair(nm_view ix, uint64_t *__restrict__ table, uint64_t first,
                                                   uint64_t n_entries, uint32_t m) {
    const uint64_t i = first + blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (i < n_entries) table[i] = nm_pair_entry<BIG>(ix, i >> 3, m, (uint32_t)(i & 7));
}

// ---- LF blocks (nm_format.h: nm_lf_entry): re-layout of the packed rank blocks, built at open ----
template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_lf_blocks(nm_view ix, nm_lf_entry *__restrict__ lfb, uint64_t n_blocks) {
    const uint64_t b = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (b >= n_blocks) return;
    nm_lf_entry e[4];
    nm_lf_entries_of_block<BIG>(ix, b, e);
#pragma unroll
    for (int c = 0; c < 4; c++) lfb[b * 4 + c] = e[c];
}

// ---- two-step rank blocks (nm_format.h: nm_rank2_block), built on the device at open -----------
// one wave per block of 64 BWT rows: row -> (c1, c2) by one LF step, six ballots give the planes,
// lanes 0..19 count the block's 16 pairs and 4 singles
template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_rank2_planes(nm_view ix, nm_rank2_block *__restrict__ r2, uint64_t n_blocks,
                                                           uint64_t *__restrict__ counts /* [20][n_blocks] */) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t b = (blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x) >> 6;
    if (b >= n_blocks) return;                              // whole waves
    const uint64_t i = b * 64 + lane;
    uint32_t c1 = 0, c2 = 0;
    bool v1 = false, v2 = false;
    if (i < ix.n) {
        v1 = nm_bwt_code(ix, i, c1);
        if (v1) v2 = nm_bwt_code(ix, nm_lf<BIG>(ix, c1, i), c2);
    }
    const uint64_t valid1 = __ballot(v1), valid2 = __ballot(v2);
    const uint64_t c1lo = __ballot(v1 && (c1 & 1u)), c1hi = __ballot(v1 && (c1 & 2u));
    const uint64_t c2lo = __ballot(v2 && (c2 & 1u)), c2hi = __ballot(v2 && (c2 & 2u));
    if (lane == 0) {
        nm_rank2_block &o = r2[b];
        o.c1lo = c1lo; o.c1hi = c1hi; o.c2lo = c2lo; o.c2hi = c2hi; o.valid1 = valid1; o.valid2 = valid2;
    }
    if (lane < 20) {
        uint64_t m;
        if (lane < 16) {
            const uint32_t x = lane >> 2, y = lane & 3;
            m = valid2 & ((x & 1u) ? c1lo : ~c1lo) & ((x & 2u) ? c1hi : ~c1hi) & ((y & 1u) ? c2lo : ~c2lo) & ((y & 2u) ? c2hi : ~c2hi);
        } else {
            const uint32_t x = lane - 16;
            m = valid1 & ((x & 1u) ? c1lo : ~c1lo) & ((x & 2u) ? c1hi : ~c1hi);
        }
        counts[(uint64_t)lane * n_blocks + b] = (uint64_t)__popcll(m);
    }
}

// counts[] now holds exclusive prefix sums per counter: make them superblock-relative and fill the
// superblock table (first row of the suffixes starting "y x" + pairs before the superblock)
template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_rank2_finish(nm_view ix, nm_rank2_block *__restrict__ r2, uint64_t n_blocks,
                                                           const uint64_t *__restrict__ counts, uint64_t *__restrict__ superC2) {
    const uint64_t t = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    const uint64_t b = t / 20;
    const uint32_t c = (uint32_t)(t % 20);
    if (b >= n_blocks) return;
    const uint64_t per_super = 1ULL << (NM_SUPER_SHIFT - 6);
    const uint64_t sb = b / per_super, b0 = sb * per_super;
    const uint64_t at_super = counts[(uint64_t)c * n_blocks + b0];
    const uint32_t rel = (uint32_t)(counts[(uint64_t)c * n_blocks + b] - at_super);
    if (c < 16) r2[b].cnt2[c] = rel; else r2[b].cnt1[c - 16] = rel;
    if (b == b0 && c < 16) {
        const uint32_t x = c >> 2, y = c & 3;
        superC2[sb * 16 + c] = nm_lf<BIG>(ix, y, ix.C[x]) + at_super;
    }
}

// ---- k_min_unique_v2: persistent waves, one lane = one position AT A TIME --------------------
// Same arithmetic as k_min_unique (nm_min_unique_one), different schedule.  In the simple kernel a
// wave runs as long as its slowest lane: with ~3 LF steps on average but a long tail, most lanes
// idle.  Here every wave owns a queue of positions (chunks of NM_CHUNK consecutive positions taken
// from a global counter); a lane that finishes its position takes the next one in the same loop
// iteration, so every iteration every lane issues exactly one dependent memory round trip:
//     SEED lane: its seed-table entry           STEP lane: the two rank blocks of lo and hi
// The chunk's encoded words (18 x 32 B) are staged in LDS, one coalesced load per chunk,
// prefetched one chunk ahead; a lane builds its 64-base window from LDS without touching HBM.
#define NM_CHUNK 1024u
#define NM_CHUNK_WORDS 18u
enum { NM_IDLE = 0, NM_SEED = 1, NM_STEP = 2 };

__device__ __forceinline__ uint64_t nm_wave_bcast64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <bool BIG, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_min_unique_v2(nm_view ix, const nm_enc_word *__restrict__ enc,
                                                            uint64_t n_enc_words, uint64_t num_kmers,
                                                            uint32_t kmin, uint32_t kmax, void *__restrict__ out,
                                                            int elem_bytes, uint64_t *__restrict__ status,
                                                            unsigned long long *__restrict__ work) {
    __shared__ nm_enc_word s_words[NM_BLOCK / NM_WAVE][NM_CHUNK_WORDS];
    const uint32_t lane = threadIdx.x & 63;
    nm_enc_word *sw = s_words[threadIdx.x >> 6];
    const uint32_t s = ix.seed_len;
    const bool use_seed = s && kmin >= s;
    const uint64_t seed_mask = (1ULL << s) - 1ULL;

    // wave-uniform queue state
    uint64_t chunk_base = 0, next_base;
    uint32_t chunk_len = 0, chunk_next = 0;
    bool more = true;
    nm_enc_word pre = {0, 0, 0, 0};
    auto grab = [&]() {
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(work, (unsigned long long)NM_CHUNK);
        next_base = nm_wave_bcast64(b);
        if (next_base < num_kmers && lane < NM_CHUNK_WORDS) {
            uint64_t w = (next_base >> 6) + lane;
            if (w >= n_enc_words) w = n_enc_words - 1;          // padding words are all-ambiguous
            pre = enc[w];
        }
    };
    grab();

    // per-lane search state
    uint32_t state = NM_IDLE, k = 0, kbase = 0;
    uint64_t p = 0, lo = 0, hi = 0, slot = 0;
    nm_window w = {0, 0, 0};
    uint32_t n_amb = 0;
    nm_tally t = {0, 0, 0, 0};
    uint32_t n_done = 0;

    for (uint32_t guard = 0; guard < (1u << 24); guard++) {      // every wave reaches an exit
        // ---- advance to the prefetched chunk when the current one is used up
        if (chunk_next >= chunk_len && more) {
            chunk_base = next_base;
            if (chunk_base >= num_kmers) { more = false; chunk_len = 0; chunk_next = 0; }
            else {
                const uint64_t left = num_kmers - chunk_base;
                chunk_len = left < NM_CHUNK ? (uint32_t)left : NM_CHUNK;
                chunk_next = 0;
                if (lane < NM_CHUNK_WORDS) sw[lane] = pre;
                __builtin_amdgcn_wave_barrier();
                grab();
            }
        }
        // ---- hand the next positions of the chunk to idle lanes
        const uint64_t idle_mask = __ballot(state == NM_IDLE);
        const uint32_t rem = chunk_len - chunk_next;
        if (idle_mask && rem) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
            if (state == NM_IDLE && rank < rem) {
                const uint32_t q = chunk_next + rank;
                p = chunk_base + q;
                w = nm_window_from(sw[q >> 6], sw[(q >> 6) + 1], q & 63);
                kbase = 0;
                if (w.amb & 1ULL) { n_amb++; nm_store(out, elem_bytes, p, 0); }
                else if (use_seed) {
                    if (w.amb & seed_mask) nm_store(out, elem_bytes, p, 0);      // U_p < s <= kmin
                    else { slot = nm_seed_slot(w, s); state = NM_SEED; if (STATS) n_done++; }
                } else { lo = 0; hi = ix.n; k = 0; state = NM_STEP; if (STATS) n_done++; }
                if (STATS && state == NM_IDLE && !(w.amb & 1ULL)) n_done++;
            }
            const uint32_t n_idle = (uint32_t)__popcll(idle_mask);
            chunk_next += n_idle < rem ? n_idle : rem;
        }
        if (!__ballot(state != NM_IDLE)) {
            if (!more && chunk_next >= chunk_len) break;
            continue;
        }
        // ---- one memory round trip per lane
        nm_blk ba = {0, 0, 0, 0, 0, 0}, bb = {0, 0, 0, 0, 0, 0};
        uint64_t e = 0;
        if (state == NM_STEP) { ba = nm_load_blk(ix, lo); bb = nm_load_blk(ix, hi); }
        else if (state == NM_SEED) e = NM_SEED_LOAD(ix, slot);
        // ---- consume it
        if (state == NM_SEED) {
            const uint32_t c = (uint32_t)(e >> NM_SEED_LO_BITS);
            if (c != NM_SEED_CNT_SAT) { lo = e & NM_SEED_LO_MASK; hi = lo + c; k = s; }
            else { lo = 0; hi = ix.n; k = 0; }
            state = NM_STEP;
            if (STATS) t.seeds++;
        } else if (state == NM_STEP) {
            const uint32_t c = 3u - nm_window_code(w, k - kbase);
            if (STATS) { t.steps++; t.blocks += ((lo >> 6) == (hi >> 6)) ? 1u : 2u; }
            lo = nm_lf_blk<BIG>(ix, c, lo, ba);
            hi = nm_lf_blk<BIG>(ix, c, hi, bb);
            k++;
        }
        // ---- decide: finished, or which base comes next
        if (state == NM_STEP) {
            const uint64_t cnt = hi - lo;
            uint32_t result = 0;
            bool done = true;
            if (cnt == 0) {                                         // search.py:699-722
                atomicMin((unsigned long long *)&status[2], (unsigned long long)p);
                atomicOr((unsigned long long *)&status[1], 1ULL);
            } else if (cnt == 1) {
                const uint32_t ans = k > kmin ? k : kmin;
                result = nm_all_valid(enc, p, w, kbase, k, ans) ? ans : 0u;
            } else if (k < kmax) {
                uint32_t j = k - kbase;
                if (j >= 64) { w = nm_load_window(enc, p + k); kbase = k; j = 0; }
                done = ((w.amb >> j) & 1ULL) != 0;                  // k == U_p and still not unique
            }
            if (done) { nm_store(out, elem_bytes, p, result); state = NM_IDLE; }
        }
    }
    // ---- wave totals
    const uint32_t amb_sum = wave_sum(n_amb);
    if (lane == 0 && amb_sum) atomicAdd((unsigned long long *)&status[0], (unsigned long long)amb_sum);
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks), c = wave_sum(t.seeds), d = wave_sum(n_done);
        if (lane == 0) {
            atomicAdd((unsigned long long *)&status[3], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)d);
        }
    }
}

template <bool BIG, bool RC, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_fixed_k(nm_view ix, const nm_enc_word *__restrict__ enc,
                                                      uint64_t seq_len, uint64_t first, uint64_t num_kmers,
                                                      const uint32_t *__restrict__ ks, uint32_t nk,
                                                      void *__restrict__ out, int elem_bytes,
                                                      uint64_t *__restrict__ status) {
    const uint64_t p = first + blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    const bool inb = p < num_kmers;
    bool amb0 = false, err = false;
    nm_tally t = {0, 0, 0, 0};
    if (inb) {
        const uint32_t r = nm_fixed_k_one<BIG, RC>(ix, enc, p, seq_len, ks, nk, amb0, err, t);
        nm_store(out, elem_bytes, p, r);
    }
    nm_epilogue<STATS>(inb, amb0, err, p, t, status);
}

template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_count(nm_view ix, const uint8_t *__restrict__ seq,
                                                    const uint64_t *__restrict__ starts,
                                                    const uint64_t *__restrict__ lens, uint64_t n,
                                                    uint32_t *__restrict__ out) {
    const uint64_t q = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (q >= n) return;
    nm_tally t = {0, 0, 0, 0};
    out[q] = nm_count_fwd_one<BIG>(ix, seq + starts[q], lens[q], t);
}

__global__ __launch_bounds__(NM_BLOCK) void k_upper(const nm_enc_word *__restrict__ enc, uint64_t num_kmers,
                                                    uint32_t kmax, uint32_t *__restrict__ out) {
    const uint64_t p = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (p < num_kmers) out[p] = nm_upper_one(enc, p, kmax);
}

// several sequences in lock-step x several indexes; nk == 0: range mode
template <bool RC>
__global__ __launch_bounds__(NM_BLOCK) void k_multi(nm_multi_args a, uint64_t seq_len, uint64_t num_kmers, uint32_t kmin,
                                                    uint32_t kmax, const uint32_t *__restrict__ ks, uint32_t nk,
                                                    void *__restrict__ out, int elem_bytes, uint64_t *__restrict__ status) {
    const uint64_t p = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    const bool inb = p < num_kmers;
    bool amb0 = false, err = false;
    nm_tally t = {0, 0, 0, 0};
    if (inb) {
        const uint32_t r = nk == 0 ? nm_min_unique_multi_one<RC>(a, p, kmin, kmax, amb0, err)
                                   : nm_fixed_k_multi_one<RC>(a, p, seq_len, ks, nk, amb0, err);
        nm_store(out, elem_bytes, p, r);
    }
    nm_epilogue<false>(inb, amb0, err, p, t, status);
}

__global__ void k_reset_status(uint64_t *__restrict__ status, unsigned long long *__restrict__ work) {
    if (threadIdx.x < NM_STATUS_WORDS) status[threadIdx.x] = threadIdx.x == 2 ? ~0ULL : 0ULL;
    if (threadIdx.x < NM_WORK_WORDS && work) work[threadIdx.x] = 0ULL;
}

// ------------------------------------------------------------------------------ host side ---

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            nm_set_error("HIP error %d (%s) at %s:%d: %s", (int)e__, hipGetErrorString(e__),  \
                         __FILE__, __LINE__, #expr);                                          \
            return NM_E_DEVICE;                                                               \
        }                                                                                     \
    } while (0)

struct nm_buffer {
    void *p = nullptr;
    uint64_t bytes = 0;
};

struct nm_index {
    int device = 0;
    nm_file_header h;
    nm_view view;
    bool big = false;
    void *d_rank = nullptr, *d_strand = nullptr, *d_sep = nullptr, *d_seed = nullptr, *d_super = nullptr;
    void *d_seed2 = nullptr;              // small secondary seed table (nm_view_for)
    uint32_t seed2_len = 0;
    void *d_quad = nullptr;               // quad table (k_min_unique_quad)
    void *d_pair = nullptr;               // pair table (k_min_unique_pair)
    void *d_rank2 = nullptr, *d_super2 = nullptr;   // two-step rank blocks + their superblock table
    void *d_lfb = nullptr;                // LF blocks
    uint64_t device_bytes = 0;
    hipStream_t stream = nullptr;
    // scratch owned by the handle (grown on demand)
    nm_buffer enc, seq, out, status, ks, starts, lens, work, settled, coarse;
    uint64_t coarse_min = 32ull << 20;    // launches of at least this many positions also run the coarse probes (NEWMAP_AMD_COARSE_MIN) ...
    int coarse_mode = 1;                  // ... 1: once an earlier launch has met long repeats, 2: always, 0: never (NEWMAP_AMD_COARSE)
    uint32_t *h_repeats_seen = nullptr;   // pinned word the fine probes set; d_repeats_seen = its device address
    uint32_t *d_repeats_seen = nullptr;
    uint32_t *d_seen_latch = nullptr;     // device-side copy of the flag
    bool list_via_range = true;           // list mode with one length runs on the range kernels (NM_OPT_LIST_VIA_RANGE, A/B)
    int quad_block = 256;                 // workgroup size of k_min_unique_quad (64 or 256, NEWMAP_AMD_QUAD_BLOCK)
    bool repeat_probes = true;            // k_repeat_probe before the both-strand range kernels (NM_OPT_REPEAT_PROBES)
    uint64_t enc_words = 0;               // words written by the last nm_encode
    int kernel_version = 0;               // 0 = automatic (pair kernel when its table exists, else 1); 1..4 force a kernel
    unsigned persistent_blocks = 2048;    // set from the device properties at open
    bool count_steps = false;
    int last_kernel = 0;                  // which range kernel the last launch used (nm_index_info 8)
    // NM_OPT_TIMING: HIP events around every search-kernel launch, on the launch stream
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;      // start/stop pairs, reused
    size_t ev_used = 0;                   // events consumed since the last read
};

struct nm_timed {                         // records start on construction, stop on destruction
    nm_index *ix; hipStream_t st; hipEvent_t stop = nullptr;
    nm_timed(nm_index *ix_, hipStream_t st_) : ix(ix_), st(st_) {
        if (!ix->timing) return;
        if (ix->ev_used + 2 > ix->ev_pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            ix->ev_pool.push_back(a); ix->ev_pool.push_back(b);
        }
        (void)hipEventRecord(ix->ev_pool[ix->ev_used], st);
        stop = ix->ev_pool[ix->ev_used + 1];
        ix->ev_used += 2;
    }
    ~nm_timed() { if (stop) (void)hipEventRecord(stop, st); }
};

static int nm_grow(nm_buffer &b, uint64_t bytes) {
    if (bytes <= b.bytes && b.p) return NM_OK;
    if (b.p) { HIP_TRY(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    uint64_t want = bytes + bytes / 8 + 4096;
    HIP_TRY(hipMalloc(&b.p, want));
    b.bytes = want;
    return NM_OK;
}

static inline unsigned nm_grid(uint64_t items) { return (unsigned)((items + NM_BLOCK - 1) / NM_BLOCK); }

extern "C" int nm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// build the 4^s table on the device; launches are sliced so that grid * block stays below 2^32
static int nm_seed_launch(nm_index *ix, const nm_view &v, const uint64_t *parent, uint64_t *table, uint32_t s) {
    const uint64_t n_slots = 1ULL << (2 * s), slice = 1ULL << 30;
    for (uint64_t first = 0; first < n_slots; first += slice) {
        const uint64_t m = n_slots - first < slice ? n_slots - first : slice;
        const dim3 grid(nm_grid(m)), block(NM_BLOCK);
        if (parent) {
            if (ix->big) hipLaunchKernelGGL(k_seed_level<true>, grid, block, 0, ix->stream, v, parent, table, first, n_slots, s);
            else         hipLaunchKernelGGL(k_seed_level<false>, grid, block, 0, ix->stream, v, parent, table, first, n_slots, s);
        } else {
            if (ix->big) hipLaunchKernelGGL(k_seed<true>, grid, block, 0, ix->stream, v, table, first, n_slots, s);
            else         hipLaunchKernelGGL(k_seed<false>, grid, block, 0, ix->stream, v, table, first, n_slots, s);
        }
        HIP_TRY(hipGetLastError());
    }
    return NM_OK;
}

// build the 4^s table on the device: level 8 entry by entry, every further level from the one
// below it (launches sliced so that grid * block stays below 2^32)
// NEWMAP_AMD_VERBOSE=1: phase timings of nm_index_open on stderr
static bool nm_verbose() { const char *v = getenv("NEWMAP_AMD_VERBOSE"); return v && *v && *v != '0'; }
static double nm_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define NM_PHASE(t0, what) do { if (nm_verbose()) { fprintf(stderr, "[open] %s: %.3fs\n", what, nm_now() - (t0)); (t0) = nm_now(); } } while (0)

// quad table for cores of m bases, from the seed table of that length (a level of the seed-table build):
// 4^m entries x 32 bytes
static int nm_build_quad(nm_index *ix, const uint64_t *level_table, uint32_t m) {
    ix->view.quad = nullptr;
    ix->view.quad_m = 0;
    if (!level_table || m < 3 || m > 16 || ix->h.n < 2) return NM_OK;
    const uint64_t n_cores = 1ULL << (2 * m);
    double tq = nm_now();
    if (hipMalloc(&ix->d_quad, n_cores * 32) != hipSuccess) {     // (someone else holds the memory: go on without the table)
        (void)hipGetLastError();
        ix->d_quad = nullptr;
        if (nm_verbose()) fprintf(stderr, "[open] quad table of %llu GB does not fit: range mode runs on the seed table\n",
                                  (unsigned long long)(n_cores * 32 >> 30));
        return NM_OK;
    }
    NM_PHASE(tq, "quad table hipMalloc");
    ix->device_bytes += n_cores * 32;
    nm_view v = ix->view;
    v.seed = level_table;
    v.seed_len = m;
    const uint64_t slice = 1ULL << 30;
    for (uint64_t first = 0; first < n_cores; first += slice) {
        const uint64_t cnt = n_cores - first < slice ? n_cores - first : slice;
        if (ix->big) hipLaunchKernelGGL(k_quad_build<true>, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, v, (uint64_t *)ix->d_quad, first, n_cores, m);
        else         hipLaunchKernelGGL(k_quad_build<false>, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, v, (uint64_t *)ix->d_quad, first, n_cores, m);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(ix->stream));
    NM_PHASE(tq, "quad table kernels");
    ix->view.quad = (const uint64_t *)ix->d_quad;
    ix->view.quad_m = m;
    return NM_OK;
}

// quad_m: also derive the quad table from the level of that length (0 = none)
static int nm_build_seed_table(nm_index *ix, uint32_t s, void **d_table, uint32_t quad_m = 0) {
    const uint64_t n_slots = 1ULL << (2 * s);
    double ts = nm_now();
    HIP_TRY(hipMalloc(d_table, n_slots * sizeof(uint64_t)));
    NM_PHASE(ts, "seed table hipMalloc");
    ix->device_bytes += n_slots * sizeof(uint64_t);
    nm_view v = ix->view;
    v.seed = nullptr;
    v.seed_len = 0;
    const uint32_t s0 = s < 8 ? s : 8;
    void *cur = nullptr;
    int rc = NM_OK;
    for (uint32_t level = s0; level <= s && rc == NM_OK; level++) {
        void *dst = *d_table;
        if (level < s && hipMalloc(&dst, (8ULL << (2 * level))) != hipSuccess) { nm_set_error("hipMalloc failed for a seed level"); rc = NM_E_ALLOC; break; }
        rc = nm_seed_launch(ix, v, level == s0 ? nullptr : (const uint64_t *)cur, (uint64_t *)dst, level);
        if (rc == NM_OK && hipStreamSynchronize(ix->stream) != hipSuccess) { nm_set_error("seed table kernel failed"); rc = NM_E_DEVICE; }
        if (rc == NM_OK && level == quad_m) rc = nm_build_quad(ix, (const uint64_t *)dst, level);
        if (cur) (void)hipFree(cur);
        cur = level < s ? dst : nullptr;
    }
    if (cur) (void)hipFree(cur);
    NM_PHASE(ts, "seed table levels (incl. the quad table)");
    return rc;
}

// core length of the quad table: as long as the seed, at most 60 % of the HBM still free once the seed table is
// in place (4^m x 32 bytes: 137 GB for m = 16; the pair table is not built next to it)
static uint32_t nm_auto_quad_len(const nm_index *ix, uint32_t s) {
    (void)ix;
    uint32_t m = s;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    free_b -= free_b < (8ULL << (2 * s)) ? free_b : (8ULL << (2 * s));      // the seed table comes first
    while (m >= 8 && (32ULL << (2 * m)) > free_b / 5 * 3) m--;
    return m >= 8 ? m : 0;
}

// seed length that makes nearly all positions resolve in the table: two more bases than log4(n)
static uint32_t nm_auto_seed_len(const nm_index *ix) {
    uint32_t s = 1;
    while (s < 16 && (1ULL << (2 * s)) < ix->h.n) s++;     // s = ceil(log4 n)
    uint32_t bonus = 2;                                   // NEWMAP_AMD_SEED_BONUS: measurement knob
    if (const char *b = getenv("NEWMAP_AMD_SEED_BONUS")) bonus = (uint32_t)atoi(b);
    s = s + bonus > 16 ? 16 : s + bonus;
    if (s < 4) s = 4;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        while (s > 4 && (8ULL << (2 * s)) > free_b / 4) s--;   // never more than a quarter of free HBM
    return s;
}

#include "nm_scan.hip.h"

// LF blocks: 64 B per 64 BWT rows (one 16-byte entry per base)
static int nm_build_lf_blocks(nm_index *ix) {
    const uint64_t n_blocks = ix->h.n / 64 + 1;
    HIP_TRY(hipMalloc(&ix->d_lfb, n_blocks * 4 * sizeof(nm_lf_entry)));
    ix->device_bytes += n_blocks * 4 * sizeof(nm_lf_entry);
    nm_view v = ix->view;
    v.lfb = nullptr;
    if (ix->big) hipLaunchKernelGGL(k_lf_blocks<true>, dim3(nm_grid(n_blocks)), dim3(NM_BLOCK), 0, ix->stream, v, (nm_lf_entry *)ix->d_lfb, n_blocks);
    else         hipLaunchKernelGGL(k_lf_blocks<false>, dim3(nm_grid(n_blocks)), dim3(NM_BLOCK), 0, ix->stream, v, (nm_lf_entry *)ix->d_lfb, n_blocks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->view.lfb = (const nm_lf_entry *)ix->d_lfb;
    return NM_OK;
}

// two-step rank blocks: 128 B per 64 BWT rows, derived on the device from the one-step structure
static int nm_build_rank2(nm_index *ix) {
    const uint64_t n_blocks = ix->h.n / 64 + 1;
    void *d_counts = nullptr, *d_scratch = nullptr, *d_total = nullptr;
    HIP_TRY(hipMalloc(&ix->d_rank2, n_blocks * sizeof(nm_rank2_block)));
    HIP_TRY(hipMalloc(&ix->d_super2, (uint64_t)ix->h.n_super * 16 * sizeof(uint64_t)));
    ix->device_bytes += n_blocks * sizeof(nm_rank2_block);
    const uint64_t scratch_n = n_blocks / TILE + n_blocks / ((uint64_t)TILE * TILE) + 8192;
    int rc = NM_OK;
    if (hipMalloc(&d_counts, 20 * n_blocks * sizeof(uint64_t)) != hipSuccess || hipMalloc(&d_scratch, scratch_n * 8) != hipSuccess ||
        hipMalloc(&d_total, 16) != hipSuccess) {
        nm_set_error("hipMalloc failed while building the two-step rank blocks");
        rc = NM_E_ALLOC;
    }
    nm_view v = ix->view;
    if (rc == NM_OK) {
        const dim3 block(NM_BLOCK);
        const unsigned g1 = nm_grid(n_blocks * 64);
        if (ix->big) hipLaunchKernelGGL(k_rank2_planes<true>, dim3(g1), block, 0, ix->stream, v, (nm_rank2_block *)ix->d_rank2, n_blocks, (uint64_t *)d_counts);
        else         hipLaunchKernelGGL(k_rank2_planes<false>, dim3(g1), block, 0, ix->stream, v, (nm_rank2_block *)ix->d_rank2, n_blocks, (uint64_t *)d_counts);
        for (int c = 0; c < 20 && rc == NM_OK; c++)
            rc = scan_exclusive((uint64_t *)d_counts + (uint64_t)c * n_blocks, n_blocks, (uint64_t *)d_scratch, (uint64_t *)d_total, ix->stream);
        if (rc == NM_OK) {
            const unsigned g2 = nm_grid(n_blocks * 20);
            if (ix->big) hipLaunchKernelGGL(k_rank2_finish<true>, dim3(g2), block, 0, ix->stream, v, (nm_rank2_block *)ix->d_rank2, n_blocks, (const uint64_t *)d_counts, (uint64_t *)ix->d_super2);
            else         hipLaunchKernelGGL(k_rank2_finish<false>, dim3(g2), block, 0, ix->stream, v, (nm_rank2_block *)ix->d_rank2, n_blocks, (const uint64_t *)d_counts, (uint64_t *)ix->d_super2);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ix->stream) != hipSuccess) { nm_set_error("building the two-step rank blocks failed"); rc = NM_E_DEVICE; }
        }
    }
    if (d_counts) (void)hipFree(d_counts);
    if (d_scratch) (void)hipFree(d_scratch);
    if (d_total) (void)hipFree(d_total);
    if (rc == NM_OK) {
        ix->view.rank2 = (const nm_rank2_block *)ix->d_rank2;
        ix->view.superC2 = (const uint64_t *)ix->d_super2;
    }
    return rc;
}

// pair table for cores of m bases: 4^m blocks x 8 entries x 8 bytes
static int nm_build_pair(nm_index *ix, uint32_t m) {
    ix->view.pair = nullptr;
    ix->view.pair_m = 0;
    if (m < 3 || ix->h.n < 2) return NM_OK;
    const uint64_t n_entries = 8ULL << (2 * m);
    HIP_TRY(hipMalloc(&ix->d_pair, n_entries * sizeof(uint64_t)));
    ix->device_bytes += n_entries * sizeof(uint64_t);
    nm_view v = ix->view;
    v.seed = nullptr;
    v.seed_len = 0;
    const uint64_t slice = 1ULL << 30;
    const bool gather = ix->view.seed && ix->view.seed_len == m + 1;      // rearrange the seed table
    for (uint64_t first = 0; first < n_entries; first += slice) {
        const uint64_t cnt = n_entries - first < slice ? n_entries - first : slice;
        if (gather) hipLaunchKernelGGL(k_pair_gather, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, ix->view.seed, (uint64_t *)ix->d_pair, first, n_entries, m);
        else if (ix->big) hipLaunchKernelGGL(k_pair<true>, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, v, (uint64_t *)ix->d_pair, first, n_entries, m);
        else         hipLaunchKernelGGL(k_pair<false>, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, v, (uint64_t *)ix->d_pair, first, n_entries, m);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->view.pair = (const uint64_t *)ix->d_pair;
    ix->view.pair_m = m;
    return NM_OK;
}

static int nm_build_seed(nm_index *ix, uint32_t s, uint32_t quad_m = 0) {
    ix->view.seed = nullptr;
    ix->view.seed_len = 0;
    if (s == 0 || ix->h.n < 2) return NM_OK;
    int rc = nm_build_seed_table(ix, s, &ix->d_seed, quad_m);
    if (rc != NM_OK) return rc;
    ix->view.seed = (const uint64_t *)ix->d_seed;
    ix->view.seed_len = s;
    return NM_OK;
}

// Range / list searches whose shortest length is below the main table's s cannot use it; they get
// a second, small table of exactly that length (built on first use, kept in the handle).
static int nm_view_for(nm_index *ix, uint32_t shortest, nm_view *v) {
    *v = ix->view;
    if (ix->view.seed_len == 0 || shortest >= ix->view.seed_len) return NM_OK;
    v->seed = nullptr;
    v->seed_len = 0;
    const uint32_t s2 = shortest > 12 ? 12 : shortest;
    if (s2 < 4) return NM_OK;
    if (ix->seed2_len != s2) {
        if (ix->d_seed2) { HIP_TRY(hipFree(ix->d_seed2)); ix->d_seed2 = nullptr; ix->device_bytes -= 8ULL << (2 * ix->seed2_len); }
        ix->seed2_len = 0;
        int rc = nm_build_seed_table(ix, s2, &ix->d_seed2);
        if (rc != NM_OK) return rc;
        ix->seed2_len = s2;
    }
    v->seed = (const uint64_t *)ix->d_seed2;
    v->seed_len = s2;
    return NM_OK;
}

extern "C" int nm_index_open(const char *index_path, int device, int seed_len_override, nm_index **out) {
    if (!index_path || !out) { nm_set_error("null argument"); return NM_E_ARGUMENT; }
    *out = nullptr;
    if (device < 0) {
        nm_set_error("device %d: this engine has no CPU path; a MI355X device index (>= 0) is required", device);
        return NM_E_DEVICE;
    }
    double t_open = nm_now();
    FILE *fp = fopen(index_path, "rb");
    if (!fp) { nm_set_error("Could not load reference index from file %s", index_path); return NM_E_FILE_OPEN; }
    nm_file_header h;
    if (fread(&h, sizeof h, 1, fp) != 1 || memcmp(h.magic, NM_MAGIC, 8) != 0 || h.version != NM_FORMAT_VERSION ||
        h.header_bytes != sizeof h) {
        fclose(fp);
        nm_set_error("%s is not a newmap_amd index (format %u): rebuild it with `newmap index`", index_path, NM_FORMAT_VERSION);
        return NM_E_FILE_FORMAT;
    }
    if (h.n_rank_blocks != h.n / 64 + 1 || h.n_strand_blocks != h.n / 64 + 1 || h.n_super != (h.n >> NM_SUPER_SHIFT) + 1 ||
        h.n_super > NM_MAX_SUPER || h.off_rank != sizeof h) {
        fclose(fp);
        nm_set_error("%s: inconsistent index header", index_path);
        return NM_E_FILE_FORMAT;
    }
    int ndev = nm_device_count();
    if (device >= ndev) {
        fclose(fp);
        nm_set_error("device %d requested but %d HIP device(s) are visible", device, ndev);
        return NM_E_DEVICE;
    }
    nm_index *ix = new (std::nothrow) nm_index();
    if (!ix) { fclose(fp); nm_set_error("out of memory"); return NM_E_ALLOC; }
    ix->device = device;
    ix->h = h;
    ix->big = h.n_super > 1;
    int rc = NM_OK;
    auto fail = [&](int code) { fclose(fp); nm_index_close(ix); return code; };
    if (hipSetDevice(device) != hipSuccess) { nm_set_error("hipSetDevice(%d) failed", device); return fail(NM_E_DEVICE); }
    if (hipStreamCreate(&ix->stream) != hipSuccess) { nm_set_error("hipStreamCreate failed"); return fail(NM_E_DEVICE); }

    const uint64_t rank_bytes = h.n_rank_blocks * sizeof(nm_rank_block);
    const uint64_t strand_bytes = h.n_strand_blocks * sizeof(nm_strand_block);
    const uint64_t sep_bytes = (h.n_sep ? h.n_sep : 1) * sizeof(uint64_t);
    std::vector<uint64_t> superC(h.n_super * 4);
    uint64_t C[4];
    C[0] = h.n_sep;
    for (int c = 1; c < 4; c++) C[c] = C[c - 1] + h.base_count[c - 1];
    for (uint64_t j = 0; j < h.n_super; j++)
        for (int c = 0; c < 4; c++) superC[j * 4 + c] = C[c] + h.super_cnt[j][c];

    // stage through a bounded host buffer: the file is read once, sequentially
    auto upload = [&](void **dptr, uint64_t off, uint64_t bytes) -> int {
        if (hipMalloc(dptr, bytes ? bytes : 8) != hipSuccess) { nm_set_error("hipMalloc of %llu bytes failed", (unsigned long long)bytes); return NM_E_ALLOC; }
        ix->device_bytes += bytes;
        if (fseeko(fp, (off_t)off, SEEK_SET) != 0) { nm_set_error("seek failed in %s", index_path); return NM_E_FILE_FORMAT; }
        const uint64_t chunk = 64ULL << 20;
        std::vector<uint8_t> buf((size_t)(bytes < chunk ? bytes : chunk));
        for (uint64_t done = 0; done < bytes;) {
            const uint64_t m = bytes - done < chunk ? bytes - done : chunk;
            if (fread(buf.data(), 1, (size_t)m, fp) != m) { nm_set_error("%s is truncated", index_path); return NM_E_FILE_FORMAT; }
            if (hipMemcpy((uint8_t *)*dptr + done, buf.data(), m, hipMemcpyHostToDevice) != hipSuccess) { nm_set_error("hipMemcpy to device failed"); return NM_E_DEVICE; }
            done += m;
        }
        return NM_OK;
    };
    if ((rc = upload(&ix->d_rank, h.off_rank, rank_bytes)) != NM_OK) return fail(rc);
    if ((rc = upload(&ix->d_strand, h.off_strand, strand_bytes)) != NM_OK) return fail(rc);
    if ((rc = upload(&ix->d_sep, h.off_sep, h.n_sep * sizeof(uint64_t))) != NM_OK) return fail(rc);
    (void)sep_bytes;
    if (hipMalloc(&ix->d_super, superC.size() * sizeof(uint64_t)) != hipSuccess ||
        hipMemcpy(ix->d_super, superC.data(), superC.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
        nm_set_error("could not upload the superblock table");
        return fail(NM_E_DEVICE);
    }
    fclose(fp);
    fp = nullptr;
    NM_PHASE(t_open, "device init + index file read + upload");

    nm_view &v = ix->view;
    v.rank = (const nm_rank_block *)ix->d_rank;
    v.strand = (const nm_strand_block *)ix->d_strand;
    v.sep = (const uint64_t *)ix->d_sep;
    v.seed = nullptr;
    v.superC = (const uint64_t *)ix->d_super;
    v.n = h.n;
    v.n_sep = h.n_sep;
    for (int c = 0; c < 4; c++) v.C[c] = C[c];
    v.seed_len = 0;
    v.n_super = (uint32_t)h.n_super;
    v.seed_policy = 0;
    v.pair_m = 0;
    v.pair = nullptr;
    v.rank2 = nullptr;
    v.superC2 = nullptr;
    v.lfb = nullptr;
    v.quad = nullptr;
    v.quad_m = 0;

    if (seed_len_override < -1 && h.n >= 2) {
        const char *off = getenv("NEWMAP_AMD_LF_BLOCKS");
        if (!(off && off[0] == '0')) {
            rc = nm_build_lf_blocks(ix);
            if (rc != NM_OK) { nm_index_close(ix); return rc; }
            NM_PHASE(t_open, "LF blocks");
        }
    }
    uint32_t s = seed_len_override == -1 ? h.seed_len
               : (seed_len_override < -1 ? nm_auto_seed_len(ix) : (uint32_t)seed_len_override);
    if (s > 16) s = 16;
    // -3: automatic with small tables (seed <= 15, quad cores <= 14: 17 GB at most).  A one-shot run never earns
    // back what the large tables cost to allocate: hipMalloc of more than ~40 GB waits 3 - 5 s for the driver to
    // clear the memory (measured), the large tables save ~1.5 ps per position.
    const bool small_tables = seed_len_override == -3;
    if (small_tables && s > 15) s = 15;
    // automatic sizing: the quad table, cut from the seed-table level of its core length.  Core length:
    // NEWMAP_AMD_QUAD_M (0 = none), else nm_auto_quad_len.
    uint32_t quad_m = 0;
    if (seed_len_override < -1 && s >= 8) {
        quad_m = nm_auto_quad_len(ix, s);
        if (small_tables && quad_m > 14) quad_m = 14;
        if (const char *q = getenv("NEWMAP_AMD_QUAD_M")) quad_m = (uint32_t)atoi(q);
        if (quad_m > s) quad_m = s;
        if (quad_m && quad_m < 8) quad_m = 8;              // the level-wise build starts at length 8
    }
    rc = nm_build_seed(ix, s, quad_m);
    if (rc != NM_OK) { nm_index_close(ix); return rc; }
    const bool have_quad = ix->view.quad != nullptr;
    if (const char *qb = getenv("NEWMAP_AMD_QUAD_BLOCK")) ix->quad_block = atoi(qb) == 64 ? 64 : 256;
    if (const char *cm = getenv("NEWMAP_AMD_COARSE_MIN")) ix->coarse_min = strtoull(cm, nullptr, 10);
    if (const char *cm = getenv("NEWMAP_AMD_COARSE")) ix->coarse_mode = atoi(cm);
    if (hipHostMalloc((void **)&ix->h_repeats_seen, 64, hipHostMallocMapped) == hipSuccess) {
        *ix->h_repeats_seen = 0;
        if (hipHostGetDevicePointer((void **)&ix->d_repeats_seen, ix->h_repeats_seen, 0) != hipSuccess ||
            hipMalloc((void **)&ix->d_seen_latch, 64) != hipSuccess || hipMemset(ix->d_seen_latch, 0, 64) != hipSuccess) {
            (void)hipGetLastError();
            ix->d_repeats_seen = nullptr;
        }
    } else {
        (void)hipGetLastError();
        ix->h_repeats_seen = nullptr;
    }
    const char *force_pair = getenv("NEWMAP_AMD_PAIR");
    if (seed_len_override < -1 && s >= 5 && (!have_quad || (force_pair && force_pair[0] == '1'))) {
        // without a quad table: the pair table (cores of s-1 bases, same resolution as the
        // seed table, twice its bytes) unless it would take more than 40 % of the free HBM
        uint32_t m = s - 1 > 15 ? 15 : s - 1;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            while (m > 4 && (64ULL << (2 * m)) > free_b * 2 / 5) m--;   // at most 40 % of what is still free
        rc = nm_build_pair(ix, m);
        if (rc != NM_OK) { nm_index_close(ix); return rc; }
    }
    if (seed_len_override < -1 && ix->h.n >= 2) {
        // opt-in (NEWMAP_AMD_TWO_STEP=1): the two-step rank blocks, 2 bytes per BWT row; measured no faster
        // than one-step walks, kept for A/B
        const char *on = getenv("NEWMAP_AMD_TWO_STEP");
        if (on && on[0] == '1') {
            rc = nm_build_rank2(ix);
            if (rc != NM_OK) { nm_index_close(ix); return rc; }
        }
    }
    rc = nm_grow(ix->status, NM_STATUS_WORDS * sizeof(uint64_t));
    if (rc == NM_OK) rc = nm_grow(ix->work, NM_WORK_WORDS * sizeof(unsigned long long));
    if (rc != NM_OK) { nm_index_close(ix); return rc; }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            ix->persistent_blocks = (unsigned)prop.multiProcessorCount * 8u;    // 8 x 256 threads = 32 waves / CU
    }
    *out = ix;
    return NM_OK;
}

extern "C" void nm_index_close(nm_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    void *ptrs[] = {ix->d_rank, ix->d_strand, ix->d_sep, ix->d_seed, ix->d_seed2, ix->d_pair, ix->d_quad, ix->d_rank2, ix->d_super2, ix->d_lfb, ix->d_super, ix->enc.p, ix->seq.p,
                    ix->out.p, ix->status.p, ix->ks.p, ix->starts.p, ix->lens.p, ix->work.p, ix->settled.p, ix->coarse.p};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (ix->h_repeats_seen) (void)hipHostFree(ix->h_repeats_seen);
    if (ix->d_seen_latch) (void)hipFree(ix->d_seen_latch);
    for (hipEvent_t e : ix->ev_pool) (void)hipEventDestroy(e);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

extern "C" uint64_t nm_index_info(const nm_index *ix, int what) {
    if (!ix) return 0;
    switch (what) {
        case 0: return ix->h.n;
        case 1: return ix->h.n_fwd;
        case 2: return ix->h.n_sep;
        case 3: return ix->h.n_records;
        case 4: return ix->h.raw_bases;
        case 5: return ix->view.seed_len;
        case 6: return ix->device_bytes;
        case 7: return ix->h.sa_ratio;
        case 8: return (uint64_t)ix->last_kernel;
        case 9: return ix->view.pair_m;
        case 10: return (uint64_t)ix->device;
        case 11: return ix->view.lfb ? 1 : 0;
        case 12: return ix->view.rank2 ? 1 : 0;
        case 13: return ix->repeat_probes ? 1 : 0;
        case 18: return ix->view.quad_m;
        case 14: case 15: case 16: case 17: {              // probe tally of the last range-mode launch
            unsigned long long v = 0;
            if (hipSetDevice(ix->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 0;
            if (hipMemcpy(&v, (const unsigned long long *)ix->work.p + 1 + (what - 14), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
            return v;
        }
        default: return 0;
    }
}

extern "C" int nm_set_option(nm_index *ix, int option, int64_t value) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (option == NM_OPT_COUNT_STEPS) { ix->count_steps = value != 0; return NM_OK; }
    if (option == NM_OPT_TIMING) { ix->timing = value != 0; ix->ev_used = 0; return NM_OK; }
    if (option == NM_OPT_LF_BLOCKS) {      // A/B: LF steps read the 16-byte LF entries (if built) or the packed blocks
        ix->view.lfb = value ? (const nm_lf_entry *)ix->d_lfb : nullptr;
        return NM_OK;
    }
    if (option == NM_OPT_TWO_STEP) {       // A/B: walks use the two-step rank blocks (if built) or the one-step ones
        ix->view.rank2 = value ? (const nm_rank2_block *)ix->d_rank2 : nullptr;
        ix->view.superC2 = value ? (const uint64_t *)ix->d_super2 : nullptr;
        return NM_OK;
    }
    if (option == NM_OPT_SEED_POLICY) {
        if (value < 0 || (value & 0xFF) > 2 || value > 0x3FF) { nm_set_error("seed policy must be 0, 1 or 2 (+ 0x100 / 0x200 timing experiments)"); return NM_E_ARGUMENT; }
        ix->view.seed_policy = (uint32_t)value;
        return NM_OK;
    }
    if (option == NM_OPT_FORCE_BIG) {      // tests: run the >2^31-position code path on a small index
        ix->big = value != 0 || ix->h.n_super > 1;
        return NM_OK;
    }
    if (option == NM_OPT_LIST_VIA_RANGE) {
        ix->list_via_range = value != 0;
        return NM_OK;
    }
    if (option == NM_OPT_REPEAT_PROBES) {
        ix->repeat_probes = value != 0;
        return NM_OK;
    }
    if (option == NM_OPT_KERNEL) {
        if (value < 0 || value > 5) { nm_set_error("kernel version must be 0..5"); return NM_E_ARGUMENT; }
        ix->kernel_version = (int)value;
        return NM_OK;
    }
    if (option == NM_OPT_PERSISTENT_BLOCKS) {
        if (value < 1 || value > 65536) { nm_set_error("persistent block count out of range"); return NM_E_ARGUMENT; }
        ix->persistent_blocks = (unsigned)value;
        return NM_OK;
    }
    nm_set_error("unknown option %d", option);
    return NM_E_ARGUMENT;
}

extern "C" int nm_timing_read(nm_index *ix, uint64_t *n_launches, double *total_ms, double *max_ms) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    HIP_TRY(hipSetDevice(ix->device));
    double total = 0.0, mx = 0.0;
    for (size_t i = 0; i + 1 < ix->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(ix->ev_pool[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ix->ev_pool[i], ix->ev_pool[i + 1]));
        total += ms;
        if (ms > mx) mx = ms;
    }
    if (n_launches) *n_launches = ix->ev_used / 2;
    if (total_ms) *total_ms = total;
    if (max_ms) *max_ms = mx;
    ix->ev_used = 0;
    return NM_OK;
}

// -------------------------------------------------------------------------- launch helpers --

// d_status != nullptr: the pass also resets the launch's status words and the handle's counters
static int nm_encode(nm_index *ix, const void *d_seq, uint64_t seq_len, hipStream_t st, uint64_t *d_status = nullptr) {
    unsigned long long *work = d_status ? (unsigned long long *)ix->work.p : nullptr;
    const uint64_t n_words = seq_len / 64 + 3;
    int rc = nm_grow(ix->enc, n_words * sizeof(nm_enc_word));
    if (rc != NM_OK) return rc;
    ix->enc_words = n_words;
    if (((uintptr_t)d_seq & 15) == 0)
        hipLaunchKernelGGL(k_encode16, dim3(nm_grid(n_words * 4)), dim3(NM_BLOCK), 0, st, (const uint8_t *)d_seq, seq_len,
                           (nm_enc_word *)ix->enc.p, n_words, d_status, work);
    else
        hipLaunchKernelGGL(k_encode, dim3(nm_grid(n_words * 64)), dim3(NM_BLOCK), 0, st, (const uint8_t *)d_seq, seq_len,
                           (nm_enc_word *)ix->enc.p, n_words, d_status, work);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

static int nm_reset_status(nm_index *ix, uint64_t *d_status, hipStream_t st) {
    hipLaunchKernelGGL(k_reset_status, dim3(1), dim3(NM_WAVE), 0, st, d_status, (unsigned long long *)ix->work.p);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

static int nm_check_segment_args(const nm_index *ix, uint64_t seq_len, uint64_t num_kmers, int elem_bytes) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (num_kmers > seq_len) { nm_set_error("num_kmers (%llu) exceeds the segment length (%llu)", (unsigned long long)num_kmers, (unsigned long long)seq_len); return NM_E_ARGUMENT; }
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4) { nm_set_error("elem_bytes must be 1, 2 or 4"); return NM_E_ARGUMENT; }
    return NM_OK;
}

// the repeat probes of a launch over `n` positions: (coarse probes for large launches,) fine probes -> ix->settled
template <bool BIG>
static int nm_launch_probes(nm_index *ix, const nm_view &view, uint64_t n, uint32_t kmax, hipStream_t st, const uint32_t **words) {
    const dim3 block(NM_BLOCK);
    const nm_enc_word *enc = (const nm_enc_word *)ix->enc.p;
    const uint64_t n_probes = (n + NM_PROBE_STRIDE - 1) / NM_PROBE_STRIDE;
    int rc = nm_grow(ix->settled, (n_probes + 1) * sizeof(uint32_t));
    if (rc != NM_OK) return rc;
    unsigned long long *tally = (unsigned long long *)ix->work.p + 1;
    const uint32_t *coarse = nullptr;
    const bool repeats_met = ix->h_repeats_seen && *(volatile uint32_t *)ix->h_repeats_seen != 0;
    if (n >= ix->coarse_min && (ix->coarse_mode == 2 || (ix->coarse_mode == 1 && repeats_met))) {
        const uint64_t n_coarse = (n + NM_COARSE_STRIDE - 1) / NM_COARSE_STRIDE;
        if ((rc = nm_grow(ix->coarse, n_coarse * sizeof(uint32_t))) != NM_OK) return rc;
        if (ix->count_steps) hipLaunchKernelGGL((k_repeat_probe_coarse<BIG, true>), dim3(nm_grid(n_coarse)), block, 0, st, view, enc, n_coarse, kmax, (uint32_t *)ix->coarse.p, tally);
        else                 hipLaunchKernelGGL((k_repeat_probe_coarse<BIG, false>), dim3(nm_grid(n_coarse)), block, 0, st, view, enc, n_coarse, kmax, (uint32_t *)ix->coarse.p, tally);
        coarse = (const uint32_t *)ix->coarse.p;
    }
    if (ix->count_steps) hipLaunchKernelGGL((k_repeat_probe<BIG, true>), dim3(nm_grid(n_probes + 1)), block, 0, st, view, enc, n_probes, kmax, (uint32_t *)ix->settled.p, tally, coarse, ix->d_repeats_seen, ix->d_seen_latch);
    else                 hipLaunchKernelGGL((k_repeat_probe<BIG, false>), dim3(nm_grid(n_probes + 1)), block, 0, st, view, enc, n_probes, kmax, (uint32_t *)ix->settled.p, tally, coarse, ix->d_repeats_seen, ix->d_seen_latch);
    *words = (const uint32_t *)ix->settled.p;
    return NM_OK;
}

template <bool BIG, bool RC>
static int launch_min_unique(nm_index *ix, const nm_view &view, uint64_t num_kmers, uint32_t kmin, uint32_t kmax, void *d_out,
                              int elem_bytes, uint64_t *d_status, hipStream_t st) {
    const dim3 block(NM_BLOCK);
    const nm_enc_word *enc = (const nm_enc_word *)ix->enc.p;
    const bool quad_kernel = RC && (ix->kernel_version == 5 || ix->kernel_version == 0) && view.quad && kmin >= view.quad_m + NM_QUAD_EXT &&
                             kmin <= NM_QUAD_MAX_KMIN;
    const bool pair_kernel = !quad_kernel && RC && (ix->kernel_version == 4 || ix->kernel_version == 0) && view.pair && kmin >= view.pair_m + 1;
    // repeat probes feed the kernels that take the probe words: the quad and pair kernels and k_min_unique
    const uint32_t *settled = nullptr;
    if (RC && ix->repeat_probes && ix->kernel_version != 2 && ix->kernel_version != 3) {   // (versions 4 and 5 fall back to 1 without their table)
        const int rc = nm_launch_probes<BIG>(ix, view, num_kmers, kmax, st, &settled);
        if (rc != NM_OK) return rc;
    }
    nm_timed timed(ix, st);
    if (RC && ix->kernel_version == 2) {
        ix->last_kernel = 2;
        // persistent grid: enough waves to fill the chip, never more than there are chunks
        const uint64_t chunks = (num_kmers + NM_CHUNK - 1) / NM_CHUNK;
        uint64_t blocks = (chunks + NM_BLOCK / NM_WAVE - 1) / (NM_BLOCK / NM_WAVE);
        if (blocks > ix->persistent_blocks) blocks = ix->persistent_blocks;
        const dim3 pgrid((unsigned)blocks);
        unsigned long long *work = (unsigned long long *)ix->work.p;
        if (ix->count_steps) hipLaunchKernelGGL((k_min_unique_v2<BIG, true>), pgrid, block, 0, st, view, enc, ix->enc_words, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, work);
        else                 hipLaunchKernelGGL((k_min_unique_v2<BIG, false>), pgrid, block, 0, st, view, enc, ix->enc_words, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, work);
        return NM_OK;
    }
    if (quad_kernel) {
        const unsigned qb = ix->quad_block == 64 ? 64u : 256u;
        const uint64_t per_block = (uint64_t)(qb / NM_WAVE) * NM_QUAD_PER_WAVE;
        const dim3 qgrid((unsigned)((num_kmers + per_block - 1) / per_block)), qblock(qb);
        ix->last_kernel = 5;
        const bool longk = kmin > 60;
#define NM_LAUNCH_QUAD(STATS_, QB_, LONG_) hipLaunchKernelGGL((k_min_unique_quad<BIG, STATS_, QB_, LONG_>), qgrid, qblock, 0, st, view, enc, ix->enc_words, \
                                                              num_kmers, kmin, kmax, d_out, elem_bytes, d_status, settled)
        if (qb == 64) {
            if (ix->count_steps) { if (longk) NM_LAUNCH_QUAD(true, 64, true); else NM_LAUNCH_QUAD(true, 64, false); }
            else                 { if (longk) NM_LAUNCH_QUAD(false, 64, true); else NM_LAUNCH_QUAD(false, 64, false); }
        } else {
            if (ix->count_steps) { if (longk) NM_LAUNCH_QUAD(true, 256, true); else NM_LAUNCH_QUAD(true, 256, false); }
            else                 { if (longk) NM_LAUNCH_QUAD(false, 256, true); else NM_LAUNCH_QUAD(false, 256, false); }
        }
#undef NM_LAUNCH_QUAD
        return NM_OK;
    }
    if (pair_kernel) {
        const uint64_t per_block = (uint64_t)NM_BLOCK * 2;
        const dim3 pgrid((unsigned)((num_kmers + per_block - 1) / per_block));
        ix->last_kernel = 4;
        if (ix->count_steps) hipLaunchKernelGGL((k_min_unique_pair<BIG, true>), pgrid, block, 0, st, view, enc, ix->enc_words, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, settled);
        else                 hipLaunchKernelGGL((k_min_unique_pair<BIG, false>), pgrid, block, 0, st, view, enc, ix->enc_words, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, settled);
        return NM_OK;
    }
    if (ix->kernel_version == 3) {
        ix->last_kernel = 3;
        const uint64_t per_block = (uint64_t)NM_BLOCK * NM_MP;
        const dim3 mgrid((unsigned)((num_kmers + per_block - 1) / per_block));
        if (ix->count_steps) hipLaunchKernelGGL((k_min_unique_mp<BIG, RC, true>), mgrid, block, 0, st, view, enc, ix->enc_words, num_kmers, kmin, kmax, d_out, elem_bytes, d_status);
        else                 hipLaunchKernelGGL((k_min_unique_mp<BIG, RC, false>), mgrid, block, 0, st, view, enc, ix->enc_words, num_kmers, kmin, kmax, d_out, elem_bytes, d_status);
        return NM_OK;
    }
    const dim3 grid(nm_grid(num_kmers));
    ix->last_kernel = 1;
    if (ix->count_steps) hipLaunchKernelGGL((k_min_unique<BIG, RC, true>), grid, block, 0, st, view, enc, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, settled);
    else                 hipLaunchKernelGGL((k_min_unique<BIG, RC, false>), grid, block, 0, st, view, enc, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, settled);
    return NM_OK;
}

extern "C" int nm_min_unique_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers,
                                         uint32_t kmin, uint32_t kmax, int use_revcomp, int elem_bytes,
                                         void *d_out, uint64_t *d_status, void *stream) {
    int rc = nm_check_segment_args(ix, seq_len, num_kmers, elem_bytes);
    if (rc != NM_OK) return rc;
    if (kmin < 1 || kmin > kmax) { nm_set_error("need 1 <= kmin <= kmax (got %u, %u)", kmin, kmax); return NM_E_ARGUMENT; }
    if ((elem_bytes == 1 && kmax > 0xFF) || (elem_bytes == 2 && kmax > 0xFFFF)) { nm_set_error("kmax %u does not fit in %d-byte elements", kmax, elem_bytes); return NM_E_ARGUMENT; }
    if (!d_status) { nm_set_error("d_status is required"); return NM_E_ARGUMENT; }
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = stream ? (hipStream_t)stream : ix->stream;
    if (num_kmers == 0) return nm_reset_status(ix, d_status, st);
    if ((rc = nm_encode(ix, d_seq, seq_len, st, d_status)) != NM_OK) return rc;
    nm_view view;
    if ((rc = nm_view_for(ix, kmin, &view)) != NM_OK) return rc;
    if (ix->big) rc = use_revcomp ? launch_min_unique<true, true>(ix, view, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st)
                                  : launch_min_unique<true, false>(ix, view, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st);
    else         rc = use_revcomp ? launch_min_unique<false, true>(ix, view, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st)
                                  : launch_min_unique<false, false>(ix, view, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st);
    if (rc != NM_OK) return rc;
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

// list mode with several lengths on the quad kernel (LIST instantiation): positions [0, head), whose longest k-mer
// lies inside the data.  The caller has checked: quad table, every length >= its window, first length <= 124.
template <bool BIG>
static int launch_list_quad(nm_index *ix, const nm_view &view, uint64_t seq_len, uint64_t head, uint32_t k_first, uint32_t k_longest,
                            const uint32_t *d_ks, uint32_t nk, void *d_out, int elem_bytes, uint64_t *d_status, hipStream_t st) {
    const nm_enc_word *enc = (const nm_enc_word *)ix->enc.p;
    const uint32_t *settled = nullptr;
    if (ix->repeat_probes) {
        const int rc = nm_launch_probes<BIG>(ix, view, head, k_longest, st, &settled);
        if (rc != NM_OK) return rc;
    }
    nm_timed timed(ix, st);
    const uint64_t per_block = (uint64_t)(256 / NM_WAVE) * NM_QUAD_PER_WAVE;
    const dim3 qgrid((unsigned)((head + per_block - 1) / per_block)), qblock(256);
    ix->last_kernel = 5;
#define NM_LAUNCH_LIST(STATS_, LONG_) hipLaunchKernelGGL((k_min_unique_quad<BIG, STATS_, 256, LONG_, true>), qgrid, qblock, 0, st, view, enc, ix->enc_words, \
                                                          head, k_first, k_longest, d_out, elem_bytes, d_status, settled, seq_len, d_ks, nk)
    if (ix->count_steps) { if (k_first > 60) NM_LAUNCH_LIST(true, true); else NM_LAUNCH_LIST(true, false); }
    else                 { if (k_first > 60) NM_LAUNCH_LIST(false, true); else NM_LAUNCH_LIST(false, false); }
#undef NM_LAUNCH_LIST
    return NM_OK;
}

template <bool BIG, bool RC>
static void launch_fixed_k(nm_index *ix, const nm_view &view, uint64_t seq_len, uint64_t first, uint64_t num_kmers, const uint32_t *d_ks, uint32_t nk,
                           void *d_out, int elem_bytes, uint64_t *d_status, hipStream_t st) {
    const dim3 grid(nm_grid(num_kmers - first)), block(NM_BLOCK);      // positions [first, num_kmers)
    const nm_enc_word *enc = (const nm_enc_word *)ix->enc.p;
    nm_timed timed(ix, st);
    if (ix->count_steps) hipLaunchKernelGGL((k_fixed_k<BIG, RC, true>), grid, block, 0, st, view, enc, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status);
    else                 hipLaunchKernelGGL((k_fixed_k<BIG, RC, false>), grid, block, 0, st, view, enc, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status);
}

extern "C" int nm_fixed_k_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers,
                                      const uint32_t *ks, uint32_t nk, int use_revcomp, int elem_bytes,
                                      void *d_out, uint64_t *d_status, void *stream) {
    int rc = nm_check_segment_args(ix, seq_len, num_kmers, elem_bytes);
    if (rc != NM_OK) return rc;
    if (!ks || nk == 0) { nm_set_error("empty k list"); return NM_E_ARGUMENT; }
    uint32_t kmax = 0;
    for (uint32_t i = 0; i < nk; i++) {
        if (ks[i] < 1) { nm_set_error("k-mer lengths must be >= 1"); return NM_E_ARGUMENT; }
        if (ks[i] > kmax) kmax = ks[i];
    }
    if ((elem_bytes == 1 && kmax > 0xFF) || (elem_bytes == 2 && kmax > 0xFFFF)) { nm_set_error("k %u does not fit in %d-byte elements", kmax, elem_bytes); return NM_E_ARGUMENT; }
    if (!d_status) { nm_set_error("d_status is required"); return NM_E_ARGUMENT; }
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = stream ? (hipStream_t)stream : ix->stream;
    if ((rc = nm_reset_status(ix, d_status, st)) != NM_OK) return rc;
    if (num_kmers == 0) return NM_OK;
    if ((rc = nm_grow(ix->ks, (uint64_t)nk * sizeof(uint32_t))) != NM_OK) return rc;
    HIP_TRY(hipMemcpyAsync(ix->ks.p, ks, (uint64_t)nk * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if ((rc = nm_encode(ix, d_seq, seq_len, st)) != NM_OK) return rc;
    const uint32_t *d_ks = (const uint32_t *)ix->ks.p;
    uint32_t kshort = ks[0];
    for (uint32_t i = 1; i < nk; i++) if (ks[i] < kshort) kshort = ks[i];
    nm_view view;
    if ((rc = nm_view_for(ix, kshort, &view)) != NM_OK) return rc;
    // ONE length K on both strands is range mode with kmin = kmax = K for every position whose K-mer lies inside
    // the data (same walk, same early stop at one occurrence, same ambiguity rule): those positions take the
    // range kernels with their tables and repeat probes; the up to K-1 positions at the end of the data, whose
    // k-mer the reference truncates (search.py:590), keep the list kernel.
    uint64_t first = 0;
    if (nk == 1 && use_revcomp && ix->list_via_range && (view.quad || view.pair) && seq_len >= ks[0]) {
        const uint64_t head = num_kmers < seq_len - ks[0] + 1 ? num_kmers : seq_len - ks[0] + 1;
        if (head) {
            rc = ix->big ? launch_min_unique<true, true>(ix, view, head, ks[0], ks[0], d_out, elem_bytes, d_status, st)
                         : launch_min_unique<false, true>(ix, view, head, ks[0], ks[0], d_out, elem_bytes, d_status, st);
            if (rc != NM_OK) return rc;
            first = head;
        }
    }
    // several lengths, all at least as long as the quad table's window: the LIST instantiation of the quad kernel
    if (nk > 1 && use_revcomp && ix->list_via_range && view.quad && (ix->kernel_version == 0 || ix->kernel_version == 5) &&
        kshort >= view.quad_m + NM_QUAD_EXT && ks[0] <= NM_QUAD_MAX_KMIN && seq_len >= kmax) {
        const uint64_t head = num_kmers < seq_len - kmax + 1 ? num_kmers : seq_len - kmax + 1;
        if (head) {
            rc = ix->big ? launch_list_quad<true>(ix, view, seq_len, head, ks[0], kmax, d_ks, nk, d_out, elem_bytes, d_status, st)
                         : launch_list_quad<false>(ix, view, seq_len, head, ks[0], kmax, d_ks, nk, d_out, elem_bytes, d_status, st);
            if (rc != NM_OK) return rc;
            first = head;
        }
    }
    if (first < num_kmers) {
        if (ix->big) { if (use_revcomp) launch_fixed_k<true, true>(ix, view, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status, st); else launch_fixed_k<true, false>(ix, view, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status, st); }
        else         { if (use_revcomp) launch_fixed_k<false, true>(ix, view, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status, st); else launch_fixed_k<false, false>(ix, view, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status, st); }
    }
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

// host-buffer wrappers ---------------------------------------------------------------------

static int nm_finish_segment(nm_index *ix, void *out, uint64_t out_bytes, uint64_t *n_ambiguous, uint64_t *bad_pos) {
    uint64_t status[NM_STATUS_WORDS];
    HIP_TRY(hipMemcpyAsync(status, ix->status.p, sizeof status, hipMemcpyDeviceToHost, ix->stream));
    if (out_bytes) HIP_TRY(hipMemcpyAsync(out, ix->out.p, out_bytes, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    if (n_ambiguous) *n_ambiguous = status[0];
    if (bad_pos) *bad_pos = status[2];
    if (status[1]) {
        nm_set_error("a generated k-mer was not found in the index (first at segment position %llu); "
                     "possibly a mismatch between the sequence and the index", (unsigned long long)status[2]);
        return NM_E_KMER_NOT_FOUND;
    }
    return NM_OK;
}

static int nm_stage_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t out_bytes) {
    int rc;
    HIP_TRY(hipSetDevice(ix->device));
    if ((rc = nm_grow(ix->seq, seq_len + 64)) != NM_OK) return rc;
    if ((rc = nm_grow(ix->out, out_bytes + 64)) != NM_OK) return rc;
    if (seq_len) HIP_TRY(hipMemcpyAsync(ix->seq.p, seq, seq_len, hipMemcpyHostToDevice, ix->stream));
    return NM_OK;
}

extern "C" int nm_min_unique_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers,
                                     uint32_t kmin, uint32_t kmax, uint32_t initial_len, int use_revcomp,
                                     int elem_bytes, void *out, uint64_t *n_ambiguous, uint64_t *bad_pos) {
    (void)initial_len;   // only shapes the reference's probe schedule (search.py:429-433), never the result
    int rc = nm_check_segment_args(ix, seq_len, num_kmers, elem_bytes);
    if (rc != NM_OK) return rc;
    if ((!seq && seq_len) || (!out && num_kmers)) { nm_set_error("null buffer"); return NM_E_ARGUMENT; }
    const uint64_t out_bytes = num_kmers * (uint64_t)elem_bytes;
    if ((rc = nm_stage_segment(ix, seq, seq_len, out_bytes)) != NM_OK) return rc;
    rc = nm_min_unique_segment_dev(ix, ix->seq.p, seq_len, num_kmers, kmin, kmax, use_revcomp, elem_bytes, ix->out.p,
                                   (uint64_t *)ix->status.p, ix->stream);
    if (rc != NM_OK) return rc;
    return nm_finish_segment(ix, out, out_bytes, n_ambiguous, bad_pos);
}

extern "C" int nm_fixed_k_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers,
                                  const uint32_t *ks, uint32_t nk, int use_revcomp, int elem_bytes, void *out,
                                  uint64_t *n_ambiguous, uint64_t *bad_pos) {
    int rc = nm_check_segment_args(ix, seq_len, num_kmers, elem_bytes);
    if (rc != NM_OK) return rc;
    if ((!seq && seq_len) || (!out && num_kmers)) { nm_set_error("null buffer"); return NM_E_ARGUMENT; }
    const uint64_t out_bytes = num_kmers * (uint64_t)elem_bytes;
    if ((rc = nm_stage_segment(ix, seq, seq_len, out_bytes)) != NM_OK) return rc;
    rc = nm_fixed_k_segment_dev(ix, ix->seq.p, seq_len, num_kmers, ks, nk, use_revcomp, elem_bytes, ix->out.p,
                                (uint64_t *)ix->status.p, ix->stream);
    if (rc != NM_OK) return rc;
    return nm_finish_segment(ix, out, out_bytes, n_ambiguous, bad_pos);
}

extern "C" int nm_upper_bound_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers,
                                      uint32_t kmax, uint32_t *out) {
    int rc = nm_check_segment_args(ix, seq_len, num_kmers, 4);
    if (rc != NM_OK) return rc;
    if (seq_len - num_kmers >= kmax && kmax) {
        // newmap/search.py:780-784 asserts the same
        nm_set_error("Excess sequence buffer length is greater than the maximum search length");
        return NM_E_ARGUMENT;
    }
    const uint64_t out_bytes = num_kmers * 4;
    if ((rc = nm_stage_segment(ix, seq, seq_len, out_bytes)) != NM_OK) return rc;
    if (num_kmers == 0) return NM_OK;
    if ((rc = nm_encode(ix, ix->seq.p, seq_len, ix->stream)) != NM_OK) return rc;
    hipLaunchKernelGGL(k_upper, dim3(nm_grid(num_kmers)), dim3(NM_BLOCK), 0, ix->stream, (const nm_enc_word *)ix->enc.p,
                       num_kmers, kmax, (uint32_t *)ix->out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, ix->out.p, out_bytes, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    return NM_OK;
}

extern "C" int nm_count_from_sequence(nm_index *ix, const uint8_t *seq, uint64_t seq_len, const uint64_t *starts,
                                      const uint64_t *lens, uint64_t n, uint32_t *counts_out) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (n == 0) return NM_OK;
    if (!seq || !starts || !lens || !counts_out) { nm_set_error("null buffer"); return NM_E_ARGUMENT; }
    for (uint64_t i = 0; i < n; i++) {
        // src/newmap-count.c:184-190 (IndexError in the wrapper)
        if (starts[i] > seq_len || lens[i] > seq_len - starts[i]) {
            nm_set_error("The sum of the index and length of each k-mer must be less than or equal to the "
                         "length of the input byte sequence (query %llu)", (unsigned long long)i);
            return NM_E_ARGUMENT;
        }
    }
    int rc;
    HIP_TRY(hipSetDevice(ix->device));
    if ((rc = nm_grow(ix->seq, seq_len + 64)) != NM_OK) return rc;
    if ((rc = nm_grow(ix->starts, n * 8)) != NM_OK) return rc;
    if ((rc = nm_grow(ix->lens, n * 8)) != NM_OK) return rc;
    if ((rc = nm_grow(ix->out, n * 4)) != NM_OK) return rc;
    hipStream_t st = ix->stream;
    HIP_TRY(hipMemcpyAsync(ix->seq.p, seq, seq_len, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ix->starts.p, starts, n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ix->lens.p, lens, n * 8, hipMemcpyHostToDevice, st));
    if (ix->big) hipLaunchKernelGGL(k_count<true>, dim3(nm_grid(n)), dim3(NM_BLOCK), 0, st, ix->view, (const uint8_t *)ix->seq.p, (const uint64_t *)ix->starts.p, (const uint64_t *)ix->lens.p, n, (uint32_t *)ix->out.p);
    else         hipLaunchKernelGGL(k_count<false>, dim3(nm_grid(n)), dim3(NM_BLOCK), 0, st, ix->view, (const uint8_t *)ix->seq.p, (const uint64_t *)ix->starts.p, (const uint64_t *)ix->lens.p, n, (uint32_t *)ix->out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(counts_out, ix->out.p, n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NM_OK;
}

extern "C" int nm_count_kmers(nm_index *ix, const uint8_t *kmers, const uint64_t *offsets, uint64_t n, uint32_t *counts_out) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (n == 0) return NM_OK;
    if (!kmers || !offsets || !counts_out) { nm_set_error("null buffer"); return NM_E_ARGUMENT; }
    std::vector<uint64_t> starts(n), lens(n);
    for (uint64_t i = 0; i < n; i++) {
        if (offsets[i + 1] <= offsets[i]) {
            // src/newmap-count.c:64-69
            nm_set_error("All elements of the kmer list must have non-zero length");
            return NM_E_ARGUMENT;
        }
        starts[i] = offsets[i] - offsets[0];
        lens[i] = offsets[i + 1] - offsets[i];
    }
    return nm_count_from_sequence(ix, kmers + offsets[0], offsets[n] - offsets[0], starts.data(), lens.data(), n, counts_out);
}

extern "C" int nm_search_segment_multi(nm_index *const *indexes, uint32_t n_indexes, const uint8_t *const *seqs,
                                       uint32_t n_seqs, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks,
                                       uint32_t nk, int range_mode, int use_revcomp, int elem_bytes, void *out,
                                       uint64_t *n_ambiguous, uint64_t *bad_pos) {
    if (!indexes || !seqs || n_indexes == 0 || n_seqs == 0 || !ks || nk == 0) { nm_set_error("null or empty argument"); return NM_E_ARGUMENT; }
    if (n_indexes > NM_MAX_MULTI || n_seqs > NM_MAX_MULTI) { nm_set_error("at most %d index files and %d FASTA files are supported", NM_MAX_MULTI, NM_MAX_MULTI); return NM_E_ARGUMENT; }
    nm_index *ix0 = indexes[0];
    int rc = nm_check_segment_args(ix0, seq_len, num_kmers, elem_bytes);
    if (rc != NM_OK) return rc;
    uint32_t kmin = ks[0], kmax = ks[0];
    for (uint32_t i = 1; i < nk; i++) { if (ks[i] < kmin) kmin = ks[i]; if (ks[i] > kmax) kmax = ks[i]; }
    if (kmin < 1) { nm_set_error("k-mer lengths must be >= 1"); return NM_E_ARGUMENT; }
    if ((elem_bytes == 1 && kmax > 0xFF) || (elem_bytes == 2 && kmax > 0xFFFF)) { nm_set_error("k %u does not fit in %d-byte elements", kmax, elem_bytes); return NM_E_ARGUMENT; }
    nm_multi_args a;
    memset(&a, 0, sizeof a);
    a.n_idx = n_indexes;
    a.n_seq = n_seqs;
    for (uint32_t f = 0; f < n_indexes; f++) {
        if (!indexes[f] || indexes[f]->device != ix0->device) { nm_set_error("all indexes must be open on the same device"); return NM_E_ARGUMENT; }
        a.view[f] = indexes[f]->view;
        a.view[f].seed = nullptr;           // the multi kernels walk from the first base
        a.view[f].seed_len = 0;
    }
    HIP_TRY(hipSetDevice(ix0->device));
    hipStream_t st = ix0->stream;
    const uint64_t n_words = seq_len / 64 + 3;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) (void)hipFree(p); };
    void *d_seq = nullptr;
    if (hipMalloc(&d_seq, seq_len + 64) != hipSuccess) { nm_set_error("hipMalloc failed"); return NM_E_ALLOC; }
    tmp.push_back(d_seq);
    for (uint32_t i = 0; i < n_seqs; i++) {
        void *d_enc = nullptr;
        if (hipMalloc(&d_enc, n_words * sizeof(nm_enc_word)) != hipSuccess) { cleanup(); nm_set_error("hipMalloc failed"); return NM_E_ALLOC; }
        tmp.push_back(d_enc);
        a.enc[i] = (const nm_enc_word *)d_enc;
        if (hipMemcpyAsync(d_seq, seqs[i], seq_len, hipMemcpyHostToDevice, st) != hipSuccess) { cleanup(); nm_set_error("copy to device failed"); return NM_E_DEVICE; }
        hipLaunchKernelGGL(k_encode16, dim3(nm_grid(n_words * 4)), dim3(NM_BLOCK), 0, st, (const uint8_t *)d_seq, seq_len, (nm_enc_word *)d_enc, n_words, (uint64_t *)nullptr, (unsigned long long *)nullptr);
    }
    const uint64_t out_bytes = num_kmers * (uint64_t)elem_bytes;
    if ((rc = nm_grow(ix0->out, out_bytes + 64)) != NM_OK || (rc = nm_grow(ix0->ks, (uint64_t)nk * 4)) != NM_OK) { cleanup(); return rc; }
    if ((rc = nm_reset_status(ix0, (uint64_t *)ix0->status.p, st)) != NM_OK) { cleanup(); return rc; }
    if (hipMemcpyAsync(ix0->ks.p, ks, (uint64_t)nk * 4, hipMemcpyHostToDevice, st) != hipSuccess) { cleanup(); nm_set_error("copy to device failed"); return NM_E_DEVICE; }
    if (num_kmers) {
        const uint32_t list_n = range_mode ? 0u : nk;
        if (use_revcomp) hipLaunchKernelGGL(k_multi<true>, dim3(nm_grid(num_kmers)), dim3(NM_BLOCK), 0, st, a, seq_len, num_kmers, kmin, kmax, (const uint32_t *)ix0->ks.p, list_n, ix0->out.p, elem_bytes, (uint64_t *)ix0->status.p);
        else             hipLaunchKernelGGL(k_multi<false>, dim3(nm_grid(num_kmers)), dim3(NM_BLOCK), 0, st, a, seq_len, num_kmers, kmin, kmax, (const uint32_t *)ix0->ks.p, list_n, ix0->out.p, elem_bytes, (uint64_t *)ix0->status.p);
        if (hipGetLastError() != hipSuccess) { cleanup(); nm_set_error("kernel launch failed"); return NM_E_DEVICE; }
    }
    rc = nm_finish_segment(ix0, out, out_bytes, n_ambiguous, bad_pos);
    cleanup();
    return rc;
}

// small device helpers ------------------------------------------------------------------------

extern "C" int nm_dev_alloc(int device, uint64_t bytes, void **out) {
    if (!out) { nm_set_error("null argument"); return NM_E_ARGUMENT; }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 8));
    return NM_OK;
}
extern "C" int nm_dev_free(int device, void *p) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(p));
    return NM_OK;
}
extern "C" int nm_dev_upload(int device, void *dst, const void *src, uint64_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return NM_OK;
}
extern "C" int nm_dev_download(int device, void *dst, const void *src, uint64_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return NM_OK;
}
extern "C" int nm_dev_sync(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    return NM_OK;
}

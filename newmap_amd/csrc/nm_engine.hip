// nm_engine.hip -- the MI355X (gfx950 / CDNA4) engine behind the C-ABI of include/newmap_amd.h.
//
// Kernels (all integer / bit work, HBM-gather bound, no MFMA):
//   k_encode16 / k_encode   sequence bytes -> 2 bit-planes + ambiguity plane, 32 B / 64 bases
//   k_sites                 range mode: ONE quad-table entry per group of kmin - m + 1 positions settles the group
//                           (nm_core.h "sites"); writes the elements and the bitmap of positions left open
//   k_repeat_probe(_coarse) one walk per 64 (512) positions where the bitmap is dense: settles long repeats,
//                           fixes the lengths between equal ends
//   k_resolve               the positions k_sites left open: probe words, else seed table + walk
//                           (the three together: newmap/search.py:383-548)
//   k_min_unique            range mode, one lane per genome position (--norc, kmin below the table's window, A/B)
//   k_fixed_k               list mode,  one lane per genome position (newmap/search.py:551-644)
//   k_multi                 several FASTA files x several index files (newmap/search.py:461, 656-697)
//   k_count                 forward-strand counts of (start, len) k-mers (src/newmap-count.c:91-206)
//   k_upper                 per-position upper search length (newmap/search.py:744-882)
//   k_seed, k_seed_level, k_quad_build, k_lf_blocks   tables built at open
// The per-position logic lives in nm_core.h.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/newmap_amd.h"
#include "nm_format.h"
#include "nm_internal.h"

extern "C" int nm_index_has_record(const nm_index *ix, uint64_t length, uint64_t hash);
#define NM_HD __device__ __forceinline__
#define NM_HASH_FN __host__ __device__ __forceinline__      /* the fingerprint helpers also run on the host (tables, joins) */
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

// ---- one quad-table entry per lane, one LINE per lane pair and load instruction --------------------------------------
// A lookup reads two 16-byte halves of its 128-byte entry (windows 0, 1 and windows 3, 4).  Two load instructions of one
// lane to the same line cost a fifth of the line rate (tools/gather_ceiling: 38 vs 48 G lines/s on a 32 GiB table -- the
// second request is a separate L1 -> L2 transaction); two LANES of one instruction that read the same line are coalesced.
// So neighbouring lanes trade halves: lane 2 j hands the address of its second half to lane 2 j + 1 and takes the address
// of that lane's first half; the first load instruction then reads both halves of lane 2 j's entry, the second both
// halves of lane 2 j + 1's, and the foreign words travel back -- six DPP moves (quad_perm [1, 0, 3, 2]) per lookup.
// Every lane of the wave must take part (go = false: no entry).  The entries of a launch are read once, at random, from
// a table far larger than the caches: non-temporal loads (+8 % lines/s on a 32 GiB table, nothing lost on a 2 GiB one;
// -DNM_QUAD_NT=0 for measurement builds).
#ifndef NM_QUAD_NT
#define NM_QUAD_NT 1
#endif
typedef unsigned long long nm_u64x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t nm_swap1(uint32_t v) {      // the value of lane ^ 1
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}
static __device__ __forceinline__ uint64_t nm_swap1_64(uint64_t v) {
    return (uint64_t)nm_swap1((uint32_t)v) | ((uint64_t)nm_swap1((uint32_t)(v >> 32)) << 32);
}
// (addresses that went through a lane swap are plain integers: name the global address space, or the loads become flat ones)
static __device__ __forceinline__ nm_u64x2 nm_quad_load16(uint64_t addr) {
    typedef const nm_u64x2 __attribute__((address_space(1))) *gptr;
    gptr q = (gptr)addr;
#if NM_QUAD_NT
    return __builtin_nontemporal_load(q);
#else
    return *q;
#endif
}
struct nm_quad_inflight { nm_u64x2 va, vb; };
// first half of a lookup: trade addresses, issue both loads (nothing waits here: a lane keeps several lookups in flight)
#ifndef NM_QUAD_PAIRED
#define NM_QUAD_PAIRED 1            /* -DNM_QUAD_PAIRED=0 (measurement builds): every lane reads both halves of its own entry */
#endif
static __device__ __forceinline__ nm_quad_inflight nm_quad_issue_paired(const uint64_t *entry, bool go, const uint32_t b[4]) {
#if !NM_QUAD_PAIRED
    nm_quad_inflight g;
    g.va = nm_u64x2{0, 0}; g.vb = nm_u64x2{0, 0};
    if (go) { g.va = nm_quad_load16((uint64_t)nm_quad_pair01(entry, b)); g.vb = nm_quad_load16((uint64_t)nm_quad_pair34(entry, b)); }
    return g;
#endif
    const bool even = (threadIdx.x & 1u) == 0;
    const uint64_t p01 = go ? (uint64_t)nm_quad_pair01(entry, b) : 0ULL, p34 = go ? (uint64_t)nm_quad_pair34(entry, b) : 0ULL;
    const uint64_t keep = even ? p01 : p34;                            // the half I load myself ...
    const uint64_t theirs = nm_swap1_64(even ? p34 : p01);             // ... and the half my neighbour wants
    const uint64_t pa = even ? keep : theirs, pb = even ? theirs : keep;   // instruction A: the even lane's line, B: the odd lane's
    nm_quad_inflight f;
    f.va = nm_u64x2{0, 0}; f.vb = nm_u64x2{0, 0};
    if (pa) f.va = nm_quad_load16(pa);
    if (pb) f.vb = nm_quad_load16(pb);
    return f;
}
// second half: the foreign words travel back.  e[0], e[1] = the pair at nm_quad_pair01(entry, b), e[2], e[3] = the pair at
// nm_quad_pair34(entry, b); zeros without an entry
static __device__ __forceinline__ void nm_quad_finish_paired(const nm_quad_inflight &f, uint64_t e[4]) {
#if !NM_QUAD_PAIRED
    e[0] = f.va.x; e[1] = f.va.y; e[2] = f.vb.x; e[3] = f.vb.y;
    return;
#endif
    const bool even = (threadIdx.x & 1u) == 0;
    const nm_u64x2 mine = even ? f.va : f.vb, foreign = even ? f.vb : f.va;
    const uint64_t f0 = nm_swap1_64(foreign.x), f1 = nm_swap1_64(foreign.y);   // my other half, loaded next door
    e[0] = even ? mine.x : f0; e[1] = even ? mine.y : f1;
    e[2] = even ? f0 : mine.x; e[3] = even ? f1 : mine.y;
}

#define NM_WAVE 64
#define NM_BLOCK 256

// ------------------------------------------------------------------------------ kernels ----

#define NM_WORK_WORDS 8             /* handle-owned counters: [1..4] probe tally, [5] NM_WORK_OPEN */
#define NM_WORK_OPEN 5              /* some block of k_sites left positions open: k_repeat_probe / k_resolve have work (tally = work + 1) */

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
    uint32_t lo, hi, amb;
    nm_encode_piece(seq, seq_len, t * 16, true, lo, hi, amb);
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
static_assert(NM_PROBE_STRIDE == 64, "one word of the need bitmap per probe stride");
// coarse[c] = positions from c * NM_COARSE_STRIDE on that one walk of <= kmax + NM_COARSE_STRIDE - 1 bases settles as 0
template <bool BIG, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_repeat_probe_coarse(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t n_coarse,
                                                                  uint32_t kmax, uint32_t *__restrict__ coarse,
                                                                  unsigned long long *__restrict__ probe_tally,
                                                                  const uint64_t *__restrict__ need, uint64_t n_need, uint32_t cstride) {
    if (need && probe_tally[NM_WORK_OPEN - 1] == 0) return;          // (after k_sites: nothing was left open, k_resolve returns at once too)
    const uint64_t c = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    nm_tally t = {0, 0, 0, 0};
    if (c < n_coarse) {
        uint32_t settled = 0, exact;
        // after k_sites (need != nullptr): only where the first fine stride is mostly open -- the start of a long repeat
        const uint64_t j0 = c * (cstride / NM_PROBE_STRIDE);
        if (!need || (j0 < n_need && nm_popc64(need[j0]) >= NM_PROBE_GATE_BITS))
            nm_repeat_probe_ex<BIG>(ix, enc, c * cstride, kmax, cstride, t, settled, exact);
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

// ---- tandem runs (nm_core.h: nm_period_of).  Stands in for the coarse probes on input that has shown long repeats:
// a stride whose stretch [P, P + cstride + kmax - 1) is u-periodic belongs to a run; the FIRST stride of a run walks
// kmax + u - 1 bases once, the others inherit (k_period_spread).  Strides outside runs get 0: the fine probes take them
// (walks of at most kmax + 63 bases instead of kmax + 511 -- the launch lasts as long as its longest chain).
template <bool BIG, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_period_runs(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t n_enc_words, uint64_t n_coarse,
                                                          uint32_t kmax, uint32_t *__restrict__ coarse,
                                                          unsigned long long *__restrict__ probe_tally,
                                                          const uint64_t *__restrict__ need, uint64_t n_need, uint32_t cstride) {
    if (need && probe_tally[NM_WORK_OPEN - 1] == 0) return;
    const uint64_t c = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    nm_tally t = {0, 0, 0, 0};
    if (c < n_coarse) {
        uint32_t word = 0;
        const uint64_t j0 = c * (cstride / NM_PROBE_STRIDE);
        if (!need || (j0 < n_need && nm_popc64(need[j0]) >= NM_PROBE_GATE_BITS)) {
            const uint32_t len = cstride + kmax - 1;
            const uint32_t u = nm_period_of(enc, n_enc_words, c * cstride, len);
            if (u) {
                const bool first = c == 0 || nm_period_of(enc, n_enc_words, (c - 1) * cstride, len) != u;
                if (first) {
                    uint32_t settled, exact;
                    nm_repeat_probe_ex<BIG>(ix, enc, c * cstride, kmax, u, t, settled, exact);
                    word = settled == u ? cstride : 0u;            // S[P .. P + kmax + u - 1) occurs twice: so does every rotation
                } else {
                    word = NM_PERIOD_INHERIT;
                }
            }
        }
        coarse[c] = word;
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

// strides inside a run take the word of the run's first stride (runs are at most a few hundred strides long; a stride
// whose predecessors are all markers up to the look-back limit stays undecided = 0)
__global__ __launch_bounds__(NM_BLOCK) void k_period_spread(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t n_coarse) {
    const uint64_t c = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (c >= n_coarse) return;
    uint32_t v = in[c];
    if (v == NM_PERIOD_INHERIT) {
        v = 0;
        for (uint64_t j = c; j-- > 0 && c - j <= 8192;) {
            const uint32_t x = in[j];
            if (x != NM_PERIOD_INHERIT) { v = x; break; }
        }
    }
    out[c] = v;
}

template <bool BIG, bool STATS>
__global__ __launch_bounds__(NM_BLOCK) void k_repeat_probe(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t n_probes,
                                                           uint32_t kmax, uint32_t *__restrict__ probe,
                                                           unsigned long long *__restrict__ probe_tally,
                                                           const uint32_t *__restrict__ coarse, volatile uint32_t *repeats_seen,
                                                           uint32_t *__restrict__ seen_latch,
                                                           const uint64_t *__restrict__ need, uint64_t n_need, uint32_t cstride) {
    if (need && probe_tally[NM_WORK_OPEN - 1] == 0) return;          // (after k_sites: nothing was left open, k_resolve returns at once too)
    const uint64_t j = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    nm_tally t = {0, 0, 0, 0};
    uint32_t c = 0;
    if (j <= n_probes) {
        uint32_t word = 0;
        if (j < n_probes) {
            const uint64_t P = j * NM_PROBE_STRIDE;
            // a stride the coarse probe settles completely: the word this probe would find after kmax + 63 steps
            if (need && !nm_probe_gate(need, j, n_need)) word = 0;      // (after k_sites: nothing open here, nothing to tell)
            else if (coarse && nm_coarse_covers(coarse[P / cstride], (uint32_t)(P % cstride), NM_PROBE_STRIDE)) word = NM_PROBE_STRIDE;
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

// ---- k_sites: one 128-byte table line serves a GROUP of 5 + d positions ---------------------------
// (nm_core.h "sites".)  Range mode on both strands with m + 4 <= kmin <= NM_SITE_MAX_KMIN, and list mode whose
// first length takes the place of kmin.  A block owns BP = 512 G consecutive positions (G = d + 5, d = kmin - m - 4
// capped at NM_SITE_MAX_D): every lane looks up the sites of two groups (both loads in flight before either is
// used), ORs the positions its entries settle into a bitmap in LDS, and the block then writes the elements four
// at a time in position order -- kmin where settled and the kmin bases are unambiguous, else 0 -- together with the
// bitmap of the positions that are still open (unambiguous over kmin bases, not settled): need[j] = positions
// 64 j .. 64 j + 63 of the segment.  k_resolve finishes those.
// The kernel starts from the raw sequence bytes: a block encodes its own stretch (plus lookahead) into LDS -- 16 bytes per
// lane and turn, bit-sliced (nm_encode_piece), four lanes make one 64-base word -- and leaves its words in the segment's
// encoded array for the kernels that may follow (repeat probes, k_resolve); there is no separate encode pass.
#define NM_SITE_BLOCK 256
#define NM_SITE_PER_LANE 2
static inline uint32_t nm_site_block_positions(uint32_t d) { return NM_SITE_BLOCK * NM_SITE_PER_LANE * (d + 5); }
static inline size_t nm_site_lds_bytes(uint32_t d, uint32_t kmax) {
    const uint32_t bp = nm_site_block_positions(d);
    return (size_t)NM_SITE_STAGE_WORDS(bp, kmax) * sizeof(nm_enc_word) + (size_t)bp / 8 * 2;
}

#define NM_SITE_WALK_MAX 64u        /* open positions a block finishes itself (seed table + walk); more: left to the probes and k_resolve */
#define NM_SITE_CHANCE_MAX 256u     /* open positions a block asks the second table about (one lane each); more: a long repeat, not worth the lines */

template <bool BIG, bool STATS, bool LIST>
__global__ __launch_bounds__(NM_SITE_BLOCK) void k_sites(nm_view ix, const uint8_t *__restrict__ seq, uint64_t seq_len,
                                                         nm_enc_word *__restrict__ enc_out, uint64_t n_enc_words,
                                                         uint64_t num_kmers, uint32_t kmin, uint32_t kmax, uint32_t d, void *__restrict__ out,
                                                         int elem_bytes, uint64_t *__restrict__ status, uint64_t *__restrict__ need,
                                                         unsigned long long *__restrict__ work,
                                                         const uint32_t *__restrict__ list, uint32_t n_list, uint64_t *__restrict__ hash_part) {
    extern __shared__ uint64_t s_mem[];
    __shared__ uint32_t s_open_total, s_qn;
    __shared__ uint32_t s_q[NM_SITE_CHANCE_MAX];
    const uint32_t G = d + 5, m = ix.quad_m;
    const uint32_t BP = NM_SITE_BLOCK * NM_SITE_PER_LANE * G;          // a multiple of 512
    const uint32_t n_stage = NM_SITE_STAGE_WORDS(BP, kmax);
    nm_enc_word *s_enc = reinterpret_cast<nm_enc_word *>(s_mem);      // words w0 .. w0 + n_stage - 1 of the segment
    uint32_t *s_set = reinterpret_cast<uint32_t *>(s_enc + n_stage);  // BP bits: settled by a site
    uint32_t *s_need = s_set + BP / 32;                               // BP bits: open
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * BP;
    const uint64_t w0 = base >> 6;
    // ---- phase 0: encode.  16 bytes per lane; lanes 4 j .. 4 j + 3 OR their pieces into word j
    const bool aligned16 = (((uintptr_t)seq) & 15u) == 0;
    for (uint32_t t = tid; t < n_stage * 4; t += NM_SITE_BLOCK) {
        uint32_t lo, hi, amb;
        nm_encode_piece(seq, seq_len, (w0 + (t >> 2)) * 64 + (t & 3) * 16, aligned16, lo, hi, amb);
        const uint32_t sub = t & 3;
        uint64_t wlo = (uint64_t)lo << (16 * sub), whi = (uint64_t)hi << (16 * sub), wamb = (uint64_t)amb << (16 * sub);
        wlo |= __shfl_xor(wlo, 1, NM_WAVE);  whi |= __shfl_xor(whi, 1, NM_WAVE);  wamb |= __shfl_xor(wamb, 1, NM_WAVE);
        wlo |= __shfl_xor(wlo, 2, NM_WAVE);  whi |= __shfl_xor(whi, 2, NM_WAVE);  wamb |= __shfl_xor(wamb, 2, NM_WAVE);
        if (sub == 0) {
            nm_enc_word w;
            w.lo = wlo; w.hi = whi; w.amb = wamb; w.pad = 0;
            s_enc[t >> 2] = w;
        }
    }
    for (uint32_t i = tid; i < BP / 16; i += NM_SITE_BLOCK) s_set[i] = 0;          // both bitmaps
    if (tid == 0) { s_open_total = 0; s_qn = 0; }
    __syncthreads();
    // the fingerprint of the block's own positions (nm_hash.h): one partial sum per block, no atomics -- tens of thousands of
    // blocks adding to ONE status word took as long as the lookups; k_resolve's first block adds the partials up
    if (hash_part) {
        __shared__ uint64_t s_hash[NM_SITE_BLOCK / NM_WAVE];
        uint64_t term = 0;
        for (uint32_t i = tid; i < BP / 64; i += NM_SITE_BLOCK) term += nm_hash_segment_word(ix.hash_tab, s_enc[i], w0 + i, num_kmers);
        if (BP / 64 > NM_WAVE) {                                       // (more words than one wave: the waves meet in LDS)
            for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, NM_WAVE);
            if ((tid & 63) == 0) s_hash[tid >> 6] = term;
            __syncthreads();
            if (tid == 0) hash_part[blockIdx.x] = s_hash[0] + s_hash[1] + s_hash[2] + s_hash[3];
        } else if (tid < NM_WAVE) {
            for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, NM_WAVE);
            if (tid == 0) hash_part[blockIdx.x] = term;
        }
    }
    // the block's own words go to the segment's encoded array; the last block also writes what follows its stretch
    // (lookahead and padding words of the segment)
    if (enc_out) {
        const uint64_t own_end = w0 + BP / 64 < n_enc_words ? w0 + BP / 64 : n_enc_words;
        const uint64_t end = blockIdx.x + 1 == gridDim.x ? n_enc_words : own_end;
        for (uint64_t wi = w0 + tid; wi < end; wi += NM_SITE_BLOCK)
            enc_out[wi] = wi - w0 < n_stage ? s_enc[wi - w0] : nm_encode_word(seq, seq_len, wi, aligned16);
    }
    auto lds_window = [&](uint32_t rel) -> nm_window {
        const uint32_t wi = rel >> 6, sh = rel & 63;
        const nm_enc_word a = s_enc[wi];
        nm_window w;
        w.lo = a.lo; w.hi = a.hi; w.amb = a.amb;
        if (sh) {
            const nm_enc_word b = s_enc[wi + 1];
            w.lo = (w.lo >> sh) | (b.lo << (64 - sh));
            w.hi = (w.hi >> sh) | (b.hi << (64 - sh));
            w.amb = (w.amb >> sh) | (b.amb << (64 - sh));
        }
        return w;
    };
    // ---- phase 1: the sites
    nm_window win[NM_SITE_PER_LANE];
    uint64_t e[NM_SITE_PER_LANE][4];                                   // one word per window of the entry
    uint32_t bidx[NM_SITE_PER_LANE][4];
    bool go[NM_SITE_PER_LANE];
    nm_quad_inflight fly[NM_SITE_PER_LANE];
    uint32_t n_entries = 0;
#pragma unroll
    for (int s = 0; s < NM_SITE_PER_LANE; s++) {
        const uint32_t g = (uint32_t)s * NM_SITE_BLOCK + tid;          // group g: positions base + g G .. + G - 1, site at + d
        win[s] = lds_window(g * G + d);
        go[s] = base + (uint64_t)g * G < num_kmers && nm_site_core_valid(win[s], m) && !(ix.seed_policy & 0x200u);
        nm_quad_index(win[s], m, bidx[s]);
        // one 128-byte line; its two 16-byte halves are read by this lane and its neighbour (nm_quad_issue_paired)
        fly[s] = nm_quad_issue_paired(ix.quad + nm_quad_slot(win[s], m) * NM_QUAD_WORDS, go[s], bidx[s]);
        if (go[s]) n_entries += 4;
    }
#pragma unroll
    for (int s = 0; s < NM_SITE_PER_LANE; s++) nm_quad_finish_paired(fly[s], e[s]);
#pragma unroll
    for (int s = 0; s < NM_SITE_PER_LANE; s++) {
        const uint64_t settled = go[s] ? nm_site_settled(nm_site_bits(win[s], m, bidx[s], e[s]), d) : 0ULL;
        if (settled) {
            const uint32_t o = ((uint32_t)s * NM_SITE_BLOCK + tid) * G;   // bit offset of the group in the block
            const uint32_t wi = o >> 5, sh = o & 31;
            atomicOr(&s_set[wi], (uint32_t)(settled << sh));
            const uint64_t rest = sh ? settled >> (32 - sh) : settled >> 16 >> 16;
            if ((uint32_t)rest) atomicOr(&s_set[wi + 1], (uint32_t)rest);
            if (rest >> 32) atomicOr(&s_set[wi + 2], (uint32_t)(rest >> 32));
        }
    }
    __syncthreads();
    // ---- phase 2: elements and open bits, four positions per lane and turn, in position order
    uint32_t n_amb = 0, n_searched = 0, n_open = 0;
    auto amb_word = [&](uint64_t i) -> uint64_t { return s_enc[i].amb; };
    const bool wide = elem_bytes == 1 && (((uintptr_t)out) & 3u) == 0;
    for (uint32_t j = tid; j < BP / 4; j += NM_SITE_BLOCK) {
        const uint32_t rel = 4 * j;
        const uint64_t q = base + rel;
        if (q >= num_kmers) break;
        const uint64_t left = num_kmers - q;
        const uint32_t inb = left >= 4 ? 0xFu : (1u << left) - 1u;
        uint32_t own_amb;
        const uint32_t valid = nm_valid4(amb_word, rel, kmin, own_amb) & inb;
        const uint32_t set4 = (s_set[rel >> 5] >> (rel & 31)) & 0xFu;
        const uint32_t hit = valid & set4, open = valid & ~set4;
        n_amb += (uint32_t)__builtin_popcount(own_amb & inb);
        n_searched += (uint32_t)__builtin_popcount(~own_amb & inb);
        if (open) { atomicOr(&s_need[rel >> 5], open << (rel & 31)); n_open += (uint32_t)__builtin_popcount(open); }
        if (wide && inb == 0xFu) {
            reinterpret_cast<uint32_t *>(out)[q >> 2] = (hit & 1u ? kmin : 0u) | (hit & 2u ? kmin << 8 : 0u) |
                                                         (hit & 4u ? kmin << 16 : 0u) | (hit & 8u ? kmin << 24 : 0u);
        } else {
#pragma unroll
            for (uint32_t t = 0; t < 4; t++)
                if ((inb >> t) & 1u) nm_store(out, elem_bytes, q + t, (hit >> t) & 1u ? kmin : 0u);
        }
    }
    if (n_open) atomicAdd(&s_open_total, n_open);
    __syncthreads();
    // the open positions of the block, one per lane: s_q[0 .. s_qn) (callers have checked that they fit)
    auto gather_open = [&]() {
        for (uint32_t i = tid; i < BP / 32; i += NM_SITE_BLOCK)
            for (uint32_t bits = s_need[i]; bits; bits &= bits - 1) s_q[atomicAdd(&s_qn, 1u)] = i * 32 + (uint32_t)__builtin_ctz(bits);
        __syncthreads();
    };
    // ---- phase 3: second chance.  A position no site settled asks the table with the longer cores (nm_second_chance; its
    // window is in LDS).  All lookups of the block are in flight together; a block with many open positions sits in a
    // long repeat and skips this.
    nm_tally t = {0, 0, 0, 0};
    if (ix.quad2 != nullptr && kmin >= ix.quad2_m + NM_QUAD_EXT && s_open_total && s_open_total <= NM_SITE_CHANCE_MAX) {
        gather_open();
        const uint32_t n_q = s_qn;
        __syncthreads();
        if (tid == 0) s_qn = 0;
        {
            // (every lane takes part in the exchange of halves, with or without a position of its own)
            const bool have = tid < n_q;
            const uint32_t rel = have ? s_q[tid] : 0u;
            const nm_window w = lds_window(rel);
            const uint32_t m2 = ix.quad2_m;
            const bool go2 = have && nm_site_core_valid(w, m2);
            uint32_t b2[4];
            uint64_t e2[4];
            nm_quad_index(w, m2, b2);
            nm_quad_finish_paired(nm_quad_issue_paired(ix.quad2 + nm_quad_slot(w, m2) * NM_QUAD_WORDS, go2, b2), e2);
            if (have) n_entries += 4;
            if (go2 && nm_second_chance_bits(ix, w, kmin, b2, e2)) {
                nm_store(out, elem_bytes, base + rel, kmin);
                atomicAnd(&s_need[rel >> 5], ~(1u << (rel & 31)));
                atomicSub(&s_open_total, 1u);
            }
        }
        __syncthreads();
    }
    // ---- phase 4: a few open positions (the rule outside long repeats): the block finishes them itself -- seed table +
    // walk -- and hands an empty bitmap on.  Many: they stay for the repeat probes and k_resolve.
    const uint32_t open_total = s_open_total;
    bool any_err = false;
    uint64_t err_pos = ~0ULL;
    // (the walks read the block's staged words -- positions relative to its first base -- so the lookahead of the
    // longest walk must have been staged: kmax <= NM_SITE_LA_MAX)
    const bool self = open_total && open_total <= NM_SITE_WALK_MAX && kmax <= NM_SITE_LA_MAX && !(ix.seed_policy & 0x100u);
    if (self) {
        gather_open();
        if (tid < s_qn) {
            const uint64_t rel = s_q[tid];
            bool amb0 = false, err = false;
            const uint32_t v = LIST ? nm_fixed_k_one<BIG, true>(ix, s_enc, rel, seq_len - base, list, n_list, amb0, err, t)
                                    : nm_min_unique_one<BIG, true>(ix, s_enc, rel, kmin, kmax, amb0, err, t);
            if (err) { any_err = true; err_pos = base + rel; }
            nm_store(out, elem_bytes, base + rel, v);
        }
    } else if (open_total && tid == 0) {
        atomicOr(&work[NM_WORK_OPEN], 1ULL);
    }
    for (uint32_t i = tid; i < BP / 64; i += NM_SITE_BLOCK)
        if (base + 64ull * i < num_kmers) need[w0 + i] = self ? 0ULL : ((uint64_t)s_need[2 * i] | ((uint64_t)s_need[2 * i + 1] << 32));

    const uint32_t amb_sum = wave_sum(n_amb);
    if ((tid & 63) == 0 && amb_sum) atomicAdd((unsigned long long *)&status[0], (unsigned long long)amb_sum);
    if (__ballot(any_err)) {
        if (any_err) atomicMin((unsigned long long *)&status[2], (unsigned long long)err_pos);
        if ((tid & 63) == 0) atomicOr((unsigned long long *)&status[1], 1ULL);
    }
    if (STATS) {
        const uint32_t c = wave_sum(n_entries), f = wave_sum(n_searched);
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks), g = wave_sum(t.seeds);
        if ((tid & 63) == 0) {
            atomicAdd((unsigned long long *)&status[5], (unsigned long long)c);           // 8-byte table words read by the sites
            atomicAdd((unsigned long long *)&status[7], (unsigned long long)f);
            if (a | b | g) {                                                                // the block's own walks
                atomicAdd((unsigned long long *)&status[3], (unsigned long long)a);
                atomicAdd((unsigned long long *)&status[4], (unsigned long long)b);
                atomicAdd((unsigned long long *)&status[6], (unsigned long long)g);
            }
        }
    }
}

// ---- k_resolve: the positions k_sites left open -----------------------------------------------------
// A block owns NM_RES_WORDS words of the need bitmap (64 positions each; one word per lane).  Scan: a lane goes
// through the set bits of its words; what the repeat probes decide (nm_probe_kstar) is stored at once, everything else
// is queued in LDS.  Walk: the queue is worked off densely by all lanes (seed table + walk, nm_min_unique_one; list
// mode: nm_fixed_k_one).  A full queue ends the scan early; it resumes after the walks.  On input without long
// repeats the bitmap is nearly empty: one scan, one short walk phase.
// LIST: list mode with several lengths (see nm_fixed_k_segment_dev): the probes only rule positions out (repeated
// over more than the longest length -> 0); every other open position goes through nm_fixed_k_one.
#define NM_RES_BLOCK 256
#define NM_RES_WORDS 256u           /* one word per lane: the walks at the end of a repeat (up to kmax steps each) run side by side, not in turns */
#define NM_RES_QCAP 2048u
template <bool BIG, bool STATS, bool LIST>
__global__ __launch_bounds__(NM_RES_BLOCK) void k_resolve(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t num_kmers,
                                                          uint32_t kmin, uint32_t kmax, void *__restrict__ out, int elem_bytes,
                                                          uint64_t *__restrict__ status, const uint64_t *__restrict__ need,
                                                          uint64_t n_need, const uint32_t *__restrict__ probe,
                                                          const unsigned long long *__restrict__ work,
                                                          uint64_t seq_len, const uint32_t *__restrict__ list, uint32_t n_list,
                                                          const uint64_t *__restrict__ hash_part, uint32_t n_hash_part) {
    if (hash_part && blockIdx.x == 0) {                    // the segment's fingerprint: the partial sums of k_sites' blocks (nm_hash.h)
        uint64_t term = 0;
        for (uint32_t i = threadIdx.x; i < n_hash_part; i += NM_RES_BLOCK) term += hash_part[i];
        for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, NM_WAVE);
        if ((threadIdx.x & 63) == 0 && term) atomicAdd((unsigned long long *)&status[NM_STATUS_HASH], (unsigned long long)term);
    }
    if (work[NM_WORK_OPEN] == 0) return;                   // every block of k_sites finished its own positions
    __shared__ uint32_t q_p[NM_RES_QCAP];
    __shared__ uint32_t q_n;
    const uint32_t tid = threadIdx.x;
    const uint64_t wbase = (uint64_t)blockIdx.x * NM_RES_WORDS;
    constexpr uint32_t PER = NM_RES_WORDS / NM_RES_BLOCK;
    static_assert(PER == 1, "one word of the bitmap per lane");
    const uint64_t my_word = wbase + tid < n_need ? need[wbase + tid] : 0ULL;
    const uint64_t any = my_word;
    if (!__syncthreads_or(any != 0)) return;
    uint32_t r = 0;                                        // words of this lane taken so far
    uint64_t bits = 0, cur = 0;                            // open bits left in the current word, its index
    uint32_t wj = 0, wj1 = 0;                              // probe words of the current stride and of the next one
    nm_tally t = {0, 0, 0, 0};
    bool any_err = false;
    uint64_t err_pos = ~0ULL;
    for (;;) {
        if (tid == 0) q_n = 0;
        __syncthreads();
        // ---- scan
        for (;;) {
            if (!bits) {
                if (r >= PER) break;
                cur = wbase + tid + (uint64_t)NM_RES_BLOCK * r;
                bits = my_word;
                r++;
                if (bits && probe) {
                    wj = probe[cur]; wj1 = probe[cur + 1];
                    const uint32_t zeros = wj & 0xFFu;     // positions repeated over more than kmax bases: element 0, as stored
                    bits &= zeros >= 64 ? 0ULL : ~((1ULL << zeros) - 1ULL);
                }
                continue;
            }
            const uint32_t o = (uint32_t)__builtin_ctzll(bits);
            const uint64_t p = cur * 64 + o;
            const uint32_t ks = probe ? nm_probe_kstar(wj, wj1, o, NM_PROBE_STRIDE, kmax) : NM_PROBE_OPEN;
            if (ks != NM_PROBE_OPEN && (!LIST || ks > kmax)) {
                // (an open position has kmin unambiguous bases.)  Range mode: the probes fixed its least unique length
                const uint32_t v = LIST ? 0u : nm_probe_element(ks, kmin, kmax, true);
                if (v) nm_store(out, elem_bytes, p, v);
                bits &= bits - 1;
                continue;
            }
            const uint32_t slot = atomicAdd(&q_n, 1u);
            if (slot >= NM_RES_QCAP) break;                // queue full: this bit waits for the next round
            q_p[slot] = (uint32_t)(p - wbase * 64);
            bits &= bits - 1;
        }
        __syncthreads();
        // ---- walk
        const uint32_t n_walk = (ix.seed_policy & 0x100u) ? 0u : (q_n < NM_RES_QCAP ? q_n : NM_RES_QCAP);   // (0x100: timing experiment, wrong results)
        for (uint32_t i = tid; i < n_walk; i += NM_RES_BLOCK) {
            const uint64_t p = wbase * 64 + q_p[i];
            bool amb0 = false, err = false;
            const uint32_t v = LIST ? nm_fixed_k_one<BIG, true>(ix, enc, p, seq_len, list, n_list, amb0, err, t)
                                    : nm_min_unique_one<BIG, true>(ix, enc, p, kmin, kmax, amb0, err, t);
            if (err) { any_err = true; if (p < err_pos) err_pos = p; }
            nm_store(out, elem_bytes, p, v);
        }
        const bool done = !bits && r >= PER;
        if (__syncthreads_and(done)) break;
    }
    (void)num_kmers;
    if (__ballot(any_err)) {
        if (any_err) atomicMin((unsigned long long *)&status[2], (unsigned long long)err_pos);
        if ((tid & 63) == 0) atomicOr((unsigned long long *)&status[1], 1ULL);
    }
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks), c = wave_sum(t.seeds);
        if ((tid & 63) == 0 && (a | b | c)) {
            atomicAdd((unsigned long long *)&status[3], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)c);           // table words read HERE ([5]: by the sites)
        }
    }
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

// the exact zero-count guard, one lane per position (nm_core.h: nm_guard_range_one / nm_guard_list_one); nk == 0: range mode
template <bool BIG, bool RC>
__global__ __launch_bounds__(NM_BLOCK) void k_guard(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t seq_len, uint64_t num_kmers,
                                                    uint32_t kmin, uint32_t kmax, uint32_t initial_len, const uint32_t *__restrict__ ks,
                                                    uint32_t nk, uint64_t *__restrict__ status) {
    const uint64_t p = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    nm_tally t = {0, 0, 0, 0};
    const bool bad = p < num_kmers && (nk ? nm_guard_list_one<BIG, RC>(ix, enc, p, seq_len, ks, nk, t)
                                          : nm_guard_range_one<BIG, RC>(ix, enc, p, kmin, kmax, initial_len, t));
    if (__ballot(bad)) {
        if (bad) atomicMin((unsigned long long *)&status[2], (unsigned long long)p);
        if ((threadIdx.x & 63) == 0) atomicOr((unsigned long long *)&status[1], 1ULL);
    }
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

// fingerprint of the positions [0, end) of a segment from its encoded words (the paths that do not run k_sites, or run it
// over a part of the positions only: list mode); status[NM_STATUS_HASH] += the sum of the words' terms
__global__ __launch_bounds__(NM_BLOCK) void k_segment_hash(const uint64_t *__restrict__ tab, const nm_enc_word *__restrict__ enc, uint64_t end,
                                                           uint64_t *__restrict__ status) {
    __shared__ uint64_t s_hash[NM_BLOCK / NM_WAVE];
    const uint64_t n_words = (end + 63) >> 6;
    uint64_t term = 0;                                     // (a fixed, small grid: a few hundred atomics on the one status word)
    for (uint64_t w = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * NM_BLOCK)
        term += nm_hash_segment_word(tab, enc[w], w, end);
    for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, NM_WAVE);
    if ((threadIdx.x & 63) == 0) s_hash[threadIdx.x >> 6] = term;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t sum = s_hash[0] + s_hash[1] + s_hash[2] + s_hash[3];
        if (sum) atomicAdd((unsigned long long *)&status[NM_STATUS_HASH], (unsigned long long)sum);
    }
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

#define NM_LANES 6
#define NM_TIMING_KINDS 5
struct nm_lane {
    hipStream_t owner = nullptr;          // the stream whose launches use this scratch
    bool ready = false;                   // side stream and events exist
    uint64_t tick = 0;                    // last use (LRU)
    hipStream_t side = nullptr;           // repeat probes of repeat-rich input run here, beside k_sites (launch_sites)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_last = nullptr;         // end of the lane's last call on its owner's stream
    nm_buffer enc, ks, work, settled, coarse, need, hashp;   // grown on demand
    uint64_t enc_words = 0;               // words written by the last nm_encode
};

struct nm_index {
    int device = 0;
    nm_file_header h;
    nm_view view;
    bool big = false;
    void *d_rank = nullptr, *d_strand = nullptr, *d_sep = nullptr, *d_seed = nullptr, *d_super = nullptr;
    void *d_seed2 = nullptr;              // small secondary seed table (nm_view_for)
    uint32_t seed2_len = 0;
    void *d_quad = nullptr;               // quad table (k_sites), cores as long as the memory allows
    void *d_quad_small = nullptr;         // a second one with shorter cores: larger groups per line on small genomes
    uint32_t quad_small_m = 0;
    void *d_lfb = nullptr;                // LF blocks
    void *d_hash_tab = nullptr;           // tables of the record fingerprint (nm_hash.h)
    std::vector<nm_record_entry> records; // (length, fingerprint) of the indexed records, sorted
    uint64_t device_bytes = 0;
    hipStream_t stream = nullptr;
    // Launch scratch comes in LANES, one per stream the caller launches on: segments given on different streams have
    // their own encoded words, bitmaps and counters and may overlap on the GPU (a 10 M-position launch leaves most of
    // the chip idle while its last blocks drain and its three small kernels run).  Lane 0 belongs to the handle's own
    // stream (host-buffer entry points); a caller stream keeps its lane until more than NM_LANES - 1 streams are in use,
    // then the least recently used lane changes hands behind its `ev_last` (nm_lane_for).
    nm_lane lanes[NM_LANES];
    nm_lane *cur = &lanes[0];             // lane of the call in progress / of the last call (nm_index_info 14..17)
    uint64_t lane_tick = 0;
    // scratch of the host-buffer entry points (they run on `stream`, one call at a time)
    nm_buffer seq, out, status, starts, lens;
    uint64_t coarse_min = 32ull << 20;    // launches of at least this many positions also run the coarse probes (NEWMAP_AMD_COARSE_MIN) ...
    uint32_t coarse_stride = NM_COARSE_STRIDE;   // positions per coarse probe (NEWMAP_AMD_COARSE_STRIDE: 128, 256, 512)
    int coarse_mode = 1;                  // ... 1: once an earlier launch has met long repeats, 2: always, 0: never (NEWMAP_AMD_COARSE)
    uint32_t *h_repeats_seen = nullptr;   // pinned word the fine probes set; d_repeats_seen = its device address
    uint32_t *d_repeats_seen = nullptr;
    uint32_t *d_seen_latch = nullptr;     // device-side copy of the flag
    bool list_via_range = true;           // list mode with one length runs on the range kernels (NM_OPT_LIST_VIA_RANGE, A/B)
    bool repeat_probes = true;            // k_repeat_probe before the both-strand range kernels (NM_OPT_REPEAT_PROBES)
    int kernel_version = 0;               // 0 = automatic (sites when the quad table applies, else 1); 1 / 5 force a kernel
    uint32_t last_site_m = 0;             // core length of the table the sites of the last launch read (nm_index_info 20)
    bool probes_beside = true;            // NEWMAP_AMD_PROBES_BESIDE=0: the probes always follow k_sites on its stream (A/B)
    int sites_blocks_per_cu = 0;          // NEWMAP_AMD_SITES_BLOCKS_PER_CU (0 = as many as fit)
    bool periodic_runs = true;            // NEWMAP_AMD_PERIODIC=0: the coarse probes walk every stride (A/B) instead of one walk per tandem run
    int site_table = 0;                   // measurement knob (NM_OPT_SITE_TABLE): 0 = pick per launch, 1 = long cores, 2 = short cores
    uint32_t site_d_cap = NM_SITE_MAX_D;  // measurement knob (NEWMAP_AMD_SITE_D): cap on d = kmin - window of the sites
    bool count_steps = false;
    int last_kernel = 0;                  // which range kernel the last launch used (nm_index_info 8)
    uint64_t guard_segments = 0;          // segments that went through the exact guard (nm_index_info 23; tests)
    bool segment_guard = true;            // NM_OPT_SEGMENT_GUARD: the host-buffer segment calls run the exact guard themselves
    uint32_t initial_len = 0;             // --initial-search-length (NM_OPT_INITIAL_LENGTH): shapes the reference's probe schedule, hence the guard
    uint64_t last_fingerprint = 0;        // status[NM_STATUS_HASH] of the last host-buffer segment call (nm_index_info 21)
    // NM_OPT_TIMING: HIP events on the launch stream, NM_TIMING_KINDS kinds of start/stop pairs:
    // kind 0 around the dominant search kernel of a segment (k_sites / k_min_unique / k_fixed_k), kind 1 around ALL the
    // kernels of the segment (encode pass, sites, probes, resolve), kinds 2 / 3 / 4 around the coarse probes, the fine
    // probes and k_resolve (each on the stream it is launched on: the probes may run on the lane's side stream)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool[NM_TIMING_KINDS];   // start/stop pairs, reused
    size_t ev_used[NM_TIMING_KINDS] = {0, 0, 0, 0, 0};  // events consumed since the last read
};

struct nm_timed {                         // records start on construction, stop on destruction
    nm_index *ix; hipStream_t st; hipEvent_t stop = nullptr;
    nm_timed(nm_index *ix_, hipStream_t st_, int kind = 0) : ix(ix_), st(st_) {
        if (!ix->timing) return;
        std::vector<hipEvent_t> &pool = ix->ev_pool[kind];
        size_t &used = ix->ev_used[kind];
        if (used + 2 > pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            pool.push_back(a); pool.push_back(b);
        }
        (void)hipEventRecord(pool[used], st);
        stop = pool[used + 1];
        used += 2;
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

// side stream, events and counters of a lane (once)
static int nm_lane_ready(nm_lane &L) {
    if (L.ready) return NM_OK;
    if (hipStreamCreateWithFlags(&L.side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();                               // (without them the probes simply follow k_sites on one stream)
        L.side = nullptr;
    }
    HIP_TRY(hipEventCreateWithFlags(&L.ev_last, hipEventDisableTiming));
    int rc = nm_grow(L.work, NM_WORK_WORDS * sizeof(unsigned long long));
    if (rc != NM_OK) return rc;
    L.ready = true;
    return NM_OK;
}

// the lane of a launch on stream `st` becomes ix->cur.  A stream keeps its lane; a new stream takes a free lane, or the
// least recently used one of lanes 1.. after waiting (on the device) for that lane's last call.
static int nm_lane_for(nm_index *ix, hipStream_t st) {
    nm_lane *pick = nullptr;
    for (nm_lane &L : ix->lanes)
        if (L.owner == st && (L.ready || &L == &ix->lanes[0])) { pick = &L; break; }
    if (!pick) {
        for (int i = 1; i < NM_LANES && !pick; i++)
            if (!ix->lanes[i].owner) pick = &ix->lanes[i];
        if (!pick) {
            pick = &ix->lanes[1];
            for (int i = 2; i < NM_LANES; i++)
                if (ix->lanes[i].tick < pick->tick) pick = &ix->lanes[i];
            HIP_TRY(hipStreamWaitEvent(st, pick->ev_last, 0));
        }
        pick->owner = st;
    }
    int rc = nm_lane_ready(*pick);
    if (rc != NM_OK) return rc;
    pick->tick = ++ix->lane_tick;
    ix->cur = pick;
    return NM_OK;
}

// a caller that is about to destroy a stream gives its lane back: the scratch stays for the next stream that needs one
extern "C" int nm_stream_release(nm_index *ix, void *stream) {
    if (!ix || !stream) return NM_OK;
    for (int i = 1; i < NM_LANES; i++) {
        nm_lane &L = ix->lanes[i];
        if (L.owner != (hipStream_t)stream) continue;
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        if (L.side) HIP_TRY(hipStreamSynchronize(L.side));
        L.owner = nullptr;
        L.tick = 0;
        if (ix->cur == &L) ix->cur = &ix->lanes[0];
    }
    return NM_OK;
}

// end of a call: whoever takes the lane over later waits for this point of the owner's stream
static int nm_lane_done(nm_index *ix, hipStream_t st) {
    if (ix->cur != &ix->lanes[0]) HIP_TRY(hipEventRecord(ix->cur->ev_last, st));
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
// 4^m entries x 128 bytes
static int nm_build_quad(nm_index *ix, const uint64_t *level_table, uint32_t m, bool small = false) {
    if (!small) { ix->view.quad = nullptr; ix->view.quad_m = 0; }
    else ix->quad_small_m = 0;
    if (!level_table || m < 3 || m > 16 || ix->h.n < 2) return NM_OK;
    const uint64_t n_cores = 1ULL << (2 * m);
    double tq = nm_now();
    void **slot = small ? &ix->d_quad_small : &ix->d_quad;
    if (hipMalloc(slot, n_cores * NM_QUAD_WORDS * 8) != hipSuccess) {     // (someone else holds the memory: go on without the table)
        (void)hipGetLastError();
        *slot = nullptr;
        if (nm_verbose()) fprintf(stderr, "[open] quad table of %llu GB does not fit: range mode runs on the seed table\n",
                                  (unsigned long long)(n_cores * NM_QUAD_WORDS * 8 >> 30));
        return NM_OK;
    }
    NM_PHASE(tq, "quad table hipMalloc");
    ix->device_bytes += n_cores * NM_QUAD_WORDS * 8;
    nm_view v = ix->view;
    v.seed = level_table;
    v.seed_len = m;
    const uint64_t slice = 1ULL << 30;
    for (uint64_t first = 0; first < n_cores; first += slice) {
        const uint64_t cnt = n_cores - first < slice ? n_cores - first : slice;
        if (ix->big) hipLaunchKernelGGL(k_quad_build<true>, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, v, (uint64_t *)*slot, first, n_cores, m);
        else         hipLaunchKernelGGL(k_quad_build<false>, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, v, (uint64_t *)*slot, first, n_cores, m);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(ix->stream));
    NM_PHASE(tq, "quad table kernels");
    if (small) ix->quad_small_m = m;
    else { ix->view.quad = (const uint64_t *)ix->d_quad; ix->view.quad_m = m; }
    return NM_OK;
}

// quad_m: also derive the quad table from the level of that length (0 = none)
static int nm_build_seed_table(nm_index *ix, uint32_t s, void **d_table, uint32_t quad_m = 0, uint32_t quad_small_m = 0) {
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
        if (rc == NM_OK && level == quad_small_m && quad_small_m != quad_m) rc = nm_build_quad(ix, (const uint64_t *)dst, level, true);
        if (cur) (void)hipFree(cur);
        cur = level < s ? dst : nullptr;
    }
    if (cur) (void)hipFree(cur);
    NM_PHASE(ts, "seed table levels (incl. the quad table)");
    return rc;
}

// core length of the quad table: as long as the seed, at most 60 % of the HBM still free once the seed table is
// in place (4^m x 128 bytes: 137 GB for m = 15)
static uint32_t nm_auto_quad_len(const nm_index *ix, uint32_t s) {
    (void)ix;
    uint32_t m = s;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    free_b -= free_b < (8ULL << (2 * s)) ? free_b : (8ULL << (2 * s));      // the seed table comes first
    while (m >= 8 && (128ULL << (2 * m)) > free_b / 5 * 3) m--;
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

static int nm_build_seed(nm_index *ix, uint32_t s, uint32_t quad_m = 0, uint32_t quad_small_m = 0) {
    ix->view.seed = nullptr;
    ix->view.seed_len = 0;
    if (s == 0 || ix->h.n < 2) return NM_OK;
    int rc = nm_build_seed_table(ix, s, &ix->d_seed, quad_m, quad_small_m);
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
    ix->lanes[0].owner = ix->stream;

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
    {   // the record list (format 2) and the fingerprint tables
        ix->records.resize(h.n_records);
        if (h.n_records && (fseeko(fp, (off_t)h.off_records, SEEK_SET) != 0 ||
                            fread(ix->records.data(), sizeof(nm_record_entry), h.n_records, fp) != h.n_records)) {
            nm_set_error("%s is truncated (record list)", index_path);
            return fail(NM_E_FILE_FORMAT);
        }
        std::sort(ix->records.begin(), ix->records.end(), [](const nm_record_entry &a, const nm_record_entry &b) {
            return a.length != b.length ? a.length < b.length : a.hash < b.hash; });
        std::vector<uint64_t> tab(NM_HASH_TAB_WORDS);
        nm_hash_fill_tables(tab.data());
        if (hipMalloc(&ix->d_hash_tab, tab.size() * sizeof(uint64_t)) != hipSuccess ||
            hipMemcpy(ix->d_hash_tab, tab.data(), tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
            nm_set_error("could not upload the fingerprint tables");
            return fail(NM_E_DEVICE);
        }
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
    v.lfb = nullptr;
    v.quad = nullptr;
    v.quad_m = 0;
    v.quad2 = nullptr;
    v.quad2_m = 0;
    v.hash_tab = (const uint64_t *)ix->d_hash_tab;

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
    // -3: automatic with small tables (seed <= 15, quad cores <= 13 + a table with shorter cores: 20 GB at most).  A one-shot run never earns
    // back what the large tables cost to allocate: hipMalloc of more than ~40 GB waits 3 - 5 s for the driver to
    // clear the memory (measured), the large tables save ~1.5 ps per position.
    const bool small_tables = seed_len_override == -3;
    if (small_tables && s > 15) s = 15;
    // automatic sizing: the quad table, cut from the seed-table level of its core length.  Core length:
    // NEWMAP_AMD_QUAD_M (0 = none), else nm_auto_quad_len.
    uint32_t quad_m = 0;
    if (seed_len_override < -1 && s >= 8) {
        quad_m = nm_auto_quad_len(ix, s);
        if (small_tables && quad_m > 13) quad_m = 13;
        if (const char *q = getenv("NEWMAP_AMD_QUAD_M")) quad_m = (uint32_t)atoi(q);
        if (quad_m > s) quad_m = s;
        if (quad_m && quad_m < 8) quad_m = 8;              // the level-wise build starts at length 8
    }
    // a second quad table with SHORT cores (larger groups per table line, nm_core.h "sites"): windows of
    // ceil(log4(20 n)) bases -- about one in twenty repeated -- when that is shorter than the first table's and the
    // table stays below 9 GB (cores <= 13; 14 on large genomes, see below); NEWMAP_AMD_QUAD_SMALL_M overrides (0 = none)
    uint32_t quad_small_m = 0;
    if (quad_m) {
        uint32_t w1 = 1;
        while (w1 < 32 && (double)(1ULL << (2 * w1)) < 20.0 * (double)h.n) w1++;
        quad_small_m = w1 > NM_QUAD_EXT + 8 ? w1 - NM_QUAD_EXT : 8;
        if (quad_small_m > 13) {                             // capped: worth its memory only while most of its windows still occur once
            auto repeated = [&](uint32_t m) { return 1.0 - exp(-(double)h.n / pow(4.0, (double)(m + NM_QUAD_EXT))); };
            quad_small_m = 13;
            if (repeated(13) > 0.15) {
                // genomes of several Gbp: cores of 14 (34 GB) for a resident handle, when the memory left after the seed
                // table and the first quad table holds that twice (3.09 Gbp, 20:200: 7 positions per line instead of 6)
                quad_small_m = 0;
                size_t free_b = 0, total_b = 0;
                const uint64_t first = (8ULL << (2 * s)) + (128ULL << (2 * quad_m)), want = 128ULL << 28;
                if (!small_tables && quad_m > 14 && repeated(14) <= 0.15 && hipMemGetInfo(&free_b, &total_b) == hipSuccess &&
                    free_b > first && want <= (free_b - first) / 2)
                    quad_small_m = 14;
            }
        }
        if (const char *q = getenv("NEWMAP_AMD_QUAD_SMALL_M")) quad_small_m = (uint32_t)atoi(q);
        if (quad_small_m && quad_small_m < 8) quad_small_m = 8;
        if (quad_small_m >= quad_m) quad_small_m = 0;
    }
    rc = nm_build_seed(ix, s, quad_m, quad_small_m);
    if (rc != NM_OK) { nm_index_close(ix); return rc; }
    if (const char *cm = getenv("NEWMAP_AMD_COARSE_MIN")) ix->coarse_min = strtoull(cm, nullptr, 10);
    if (const char *cm = getenv("NEWMAP_AMD_COARSE")) ix->coarse_mode = atoi(cm);
    if (const char *cs = getenv("NEWMAP_AMD_COARSE_STRIDE")) { const int v = atoi(cs); if (v == 128 || v == 256 || v == 512) ix->coarse_stride = (uint32_t)v; }
    if (const char *pb = getenv("NEWMAP_AMD_PROBES_BESIDE")) ix->probes_beside = pb[0] != '0';
    if (const char *pr = getenv("NEWMAP_AMD_PERIODIC")) ix->periodic_runs = pr[0] != '0';
    if (const char *sb = getenv("NEWMAP_AMD_SITES_BLOCKS_PER_CU")) ix->sites_blocks_per_cu = atoi(sb);
    if (const char *sd = getenv("NEWMAP_AMD_SITE_D")) { ix->site_d_cap = (uint32_t)atoi(sd); if (ix->site_d_cap > NM_SITE_MAX_D) ix->site_d_cap = NM_SITE_MAX_D; }
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
    rc = nm_grow(ix->status, NM_STATUS_WORDS * sizeof(uint64_t));
    if (rc == NM_OK) rc = nm_lane_ready(ix->lanes[0]);
    if (rc != NM_OK) { nm_index_close(ix); return rc; }
    *out = ix;
    return NM_OK;
}

extern "C" void nm_index_close(nm_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    (void)hipDeviceSynchronize();                              // launches on caller streams and side streams included
    void *ptrs[] = {ix->d_rank, ix->d_strand, ix->d_sep, ix->d_seed, ix->d_seed2, ix->d_quad, ix->d_quad_small, ix->d_lfb, ix->d_super, ix->d_hash_tab, ix->seq.p,
                    ix->out.p, ix->status.p, ix->starts.p, ix->lens.p};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (nm_lane &L : ix->lanes) {
        for (void *p : {L.enc.p, L.ks.p, L.work.p, L.settled.p, L.coarse.p, L.need.p, L.hashp.p})
            if (p) (void)hipFree(p);
        if (L.side) (void)hipStreamDestroy(L.side);
        for (hipEvent_t e : {L.ev_fork, L.ev_join, L.ev_last})
            if (e) (void)hipEventDestroy(e);
    }
    if (ix->h_repeats_seen) (void)hipHostFree(ix->h_repeats_seen);
    if (ix->d_seen_latch) (void)hipFree(ix->d_seen_latch);
    for (auto &pool : ix->ev_pool) for (hipEvent_t e : pool) (void)hipEventDestroy(e);
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
        case 9: return 0;                                  // (pair table: removed)
        case 10: return (uint64_t)ix->device;
        case 11: return ix->view.lfb ? 1 : 0;
        case 12: return 0;                                 // (two-step rank blocks: removed)
        case 13: return ix->repeat_probes ? 1 : 0;
        case 18: return ix->view.quad_m;
        case 19: return ix->quad_small_m;
        case 20: return ix->last_site_m;
        case 21: return ix->last_fingerprint;
        case 22: return ix->initial_len;
        case 23: return ix->guard_segments;
        case 14: case 15: case 16: case 17: {              // probe tally of the last range-mode launch
            unsigned long long v = 0;
            if (hipSetDevice(ix->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 0;
            if (hipMemcpy(&v, (const unsigned long long *)ix->cur->work.p + 1 + (what - 14), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
            return v;
        }
        default: return 0;
    }
}

extern "C" int nm_set_option(nm_index *ix, int option, int64_t value) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (option == NM_OPT_COUNT_STEPS) { ix->count_steps = value != 0; return NM_OK; }
    if (option == NM_OPT_TIMING) { ix->timing = value != 0; for (size_t &u : ix->ev_used) u = 0; return NM_OK; }
    if (option == NM_OPT_LF_BLOCKS) {      // A/B: LF steps read the 16-byte LF entries (if built) or the packed blocks
        ix->view.lfb = value ? (const nm_lf_entry *)ix->d_lfb : nullptr;
        return NM_OK;
    }
    if (option == NM_OPT_SEED_POLICY) {
        if (value < 0 || (value & 0xFF) > 2 || value > 0x7FF) { nm_set_error("seed policy must be 0, 1 or 2 (+ 0x100 / 0x200 timing experiments)"); return NM_E_ARGUMENT; }
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
    if (option == NM_OPT_SITE_D) {         // measurement / tests: cap on d = kmin - window of the sites (a group = d + 5 positions)
        if (value < 0 || value > (int64_t)NM_SITE_MAX_D) { nm_set_error("site d cap must be 0..%u", NM_SITE_MAX_D); return NM_E_ARGUMENT; }
        ix->site_d_cap = (uint32_t)value;
        return NM_OK;
    }
    if (option == NM_OPT_SITE_TABLE) {
        if (value < 0 || value > 2) { nm_set_error("site table must be 0 (pick per launch), 1 (long cores) or 2 (short cores)"); return NM_E_ARGUMENT; }
        ix->site_table = (int)value;
        return NM_OK;
    }
    if (option == NM_OPT_SEGMENT_GUARD) { ix->segment_guard = value != 0; return NM_OK; }
    if (option == NM_OPT_INITIAL_LENGTH) {
        if (value < 0 || value > 0xFFFFFFFFLL) { nm_set_error("initial search length out of range"); return NM_E_ARGUMENT; }
        ix->initial_len = (uint32_t)value;
        return NM_OK;
    }
    if (option == NM_OPT_KERNEL) {
        if (value != 0 && value != 1 && value != 5) { nm_set_error("kernel version must be 0 (automatic), 1 (one lane per position) or 5 (sites)"); return NM_E_ARGUMENT; }
        ix->kernel_version = (int)value;
        return NM_OK;
    }
    nm_set_error("unknown option %d", option);
    return NM_E_ARGUMENT;
}

extern "C" int nm_timing_read_kind(nm_index *ix, int kind, uint64_t *n_launches, double *total_ms, double *max_ms) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (kind < 0 || kind >= NM_TIMING_KINDS) { nm_set_error("timing kind must be 0 (dominant kernel), 1 (all kernels of a segment), 2 (coarse probes), 3 (fine probes) or 4 (k_resolve)"); return NM_E_ARGUMENT; }
    HIP_TRY(hipSetDevice(ix->device));
    double total = 0.0, mx = 0.0;
    std::vector<hipEvent_t> &pool = ix->ev_pool[kind];
    for (size_t i = 0; i + 1 < ix->ev_used[kind]; i += 2) {
        HIP_TRY(hipEventSynchronize(pool[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pool[i], pool[i + 1]));
        total += ms;
        if (ms > mx) mx = ms;
    }
    if (n_launches) *n_launches = ix->ev_used[kind] / 2;
    if (total_ms) *total_ms = total;
    if (max_ms) *max_ms = mx;
    ix->ev_used[kind] = 0;
    return NM_OK;
}

extern "C" int nm_timing_read(nm_index *ix, uint64_t *n_launches, double *total_ms, double *max_ms) {
    return nm_timing_read_kind(ix, 0, n_launches, total_ms, max_ms);
}

// -------------------------------------------------------------------------- launch helpers --

// d_status != nullptr: the pass also resets the launch's status words and the handle's counters
// room for the encoded words of a segment (filled by the encode pass or by k_sites)
static int nm_prepare_enc(nm_index *ix, uint64_t seq_len) {
    const uint64_t n_words = seq_len / 64 + 3;
    int rc = nm_grow(ix->cur->enc, n_words * sizeof(nm_enc_word));
    if (rc != NM_OK) return rc;
    ix->cur->enc_words = n_words;
    return NM_OK;
}

static int nm_encode(nm_index *ix, const void *d_seq, uint64_t seq_len, hipStream_t st, uint64_t *d_status = nullptr) {
    unsigned long long *work = d_status ? (unsigned long long *)ix->cur->work.p : nullptr;
    int rc = nm_prepare_enc(ix, seq_len);
    if (rc != NM_OK) return rc;
    const uint64_t n_words = ix->cur->enc_words;
    if (((uintptr_t)d_seq & 15) == 0)
        hipLaunchKernelGGL(k_encode16, dim3(nm_grid(n_words * 4)), dim3(NM_BLOCK), 0, st, (const uint8_t *)d_seq, seq_len,
                           (nm_enc_word *)ix->cur->enc.p, n_words, d_status, work);
    else
        hipLaunchKernelGGL(k_encode, dim3(nm_grid(n_words * 64)), dim3(NM_BLOCK), 0, st, (const uint8_t *)d_seq, seq_len,
                           (nm_enc_word *)ix->cur->enc.p, n_words, d_status, work);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

// status[NM_STATUS_HASH] += fingerprint of positions [0, end) of the segment whose encoded words the lane holds
static int nm_hash_positions(nm_index *ix, uint64_t end, uint64_t *d_status, hipStream_t st) {
    if (end == 0) return NM_OK;
    const uint64_t want = nm_grid((end + 63) >> 6);
    hipLaunchKernelGGL(k_segment_hash, dim3((unsigned)(want < 512 ? want : 512)), dim3(NM_BLOCK), 0, st, (const uint64_t *)ix->d_hash_tab,
                       (const nm_enc_word *)ix->cur->enc.p, end, d_status);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

static int nm_reset_status(nm_index *ix, uint64_t *d_status, hipStream_t st, bool counters = true) {
    hipLaunchKernelGGL(k_reset_status, dim3(1), dim3(NM_WAVE), 0, st, d_status, counters ? (unsigned long long *)ix->cur->work.p : nullptr);
    HIP_TRY(hipGetLastError());
    return NM_OK;
}

static int nm_check_segment_args(const nm_index *ix, uint64_t seq_len, uint64_t num_kmers, int elem_bytes) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (num_kmers > seq_len) { nm_set_error("num_kmers (%llu) exceeds the segment length (%llu)", (unsigned long long)num_kmers, (unsigned long long)seq_len); return NM_E_ARGUMENT; }
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4) { nm_set_error("elem_bytes must be 1, 2 or 4"); return NM_E_ARGUMENT; }
    return NM_OK;
}

// the repeat probes of a launch over `n` positions: (coarse probes for large launches,) fine probes -> ix->settled.
// need != nullptr (after k_sites): only the strides whose positions are mostly open get a probe.
template <bool BIG>
static int nm_launch_probes(nm_index *ix, const nm_view &view, uint64_t n, uint32_t kmax, hipStream_t st, const uint32_t **words,
                            const uint64_t *need = nullptr) {
    const dim3 block(NM_BLOCK);
    const nm_enc_word *enc = (const nm_enc_word *)ix->cur->enc.p;
    const uint64_t n_probes = (n + NM_PROBE_STRIDE - 1) / NM_PROBE_STRIDE;
    int rc = nm_grow(ix->cur->settled, (n_probes + 1) * sizeof(uint32_t));
    if (rc != NM_OK) return rc;
    unsigned long long *tally = (unsigned long long *)ix->cur->work.p + 1;
    const uint32_t *coarse = nullptr;
    const bool repeats_met = ix->h_repeats_seen && *(volatile uint32_t *)ix->h_repeats_seen != 0;
    if (n >= ix->coarse_min && (ix->coarse_mode == 2 || (ix->coarse_mode == 1 && repeats_met))) {
        const uint32_t cstride = ix->coarse_stride;
        const uint64_t n_coarse = (n + cstride - 1) / cstride;
        if ((rc = nm_grow(ix->cur->coarse, 2 * n_coarse * sizeof(uint32_t))) != NM_OK) return rc;
        nm_timed timed(ix, st, 2);
        uint32_t *c0 = (uint32_t *)ix->cur->coarse.p, *c1 = c0 + n_coarse;
        if (ix->periodic_runs) {
            if (ix->count_steps) hipLaunchKernelGGL((k_period_runs<BIG, true>), dim3(nm_grid(n_coarse)), block, 0, st, view, enc, ix->cur->enc_words, n_coarse, kmax, c1, tally, need, n_probes, cstride);
            else                 hipLaunchKernelGGL((k_period_runs<BIG, false>), dim3(nm_grid(n_coarse)), block, 0, st, view, enc, ix->cur->enc_words, n_coarse, kmax, c1, tally, need, n_probes, cstride);
            hipLaunchKernelGGL(k_period_spread, dim3(nm_grid(n_coarse)), block, 0, st, (const uint32_t *)c1, c0, n_coarse);
        } else {
            if (ix->count_steps) hipLaunchKernelGGL((k_repeat_probe_coarse<BIG, true>), dim3(nm_grid(n_coarse)), block, 0, st, view, enc, n_coarse, kmax, c0, tally, need, n_probes, cstride);
            else                 hipLaunchKernelGGL((k_repeat_probe_coarse<BIG, false>), dim3(nm_grid(n_coarse)), block, 0, st, view, enc, n_coarse, kmax, c0, tally, need, n_probes, cstride);
        }
        coarse = (const uint32_t *)c0;
    }
    {
        nm_timed timed(ix, st, 3);
        if (ix->count_steps) hipLaunchKernelGGL((k_repeat_probe<BIG, true>), dim3(nm_grid(n_probes + 1)), block, 0, st, view, enc, n_probes, kmax, (uint32_t *)ix->cur->settled.p, tally, coarse, ix->d_repeats_seen, ix->d_seen_latch, need, n_probes, ix->coarse_stride);
        else                 hipLaunchKernelGGL((k_repeat_probe<BIG, false>), dim3(nm_grid(n_probes + 1)), block, 0, st, view, enc, n_probes, kmax, (uint32_t *)ix->cur->settled.p, tally, coarse, ix->d_repeats_seen, ix->d_seen_latch, need, n_probes, ix->coarse_stride);
    }
    *words = (const uint32_t *)ix->cur->settled.p;
    return NM_OK;
}

// can the sites (k_sites + k_resolve) take a both-strand search whose shortest length is kmin?
static bool nm_sites_apply(const nm_index *ix, const nm_view &view, uint32_t kmin) {
    if (!((ix->kernel_version == 0 || ix->kernel_version == 5) && view.quad && kmin <= NM_SITE_MAX_KMIN)) return false;
    return kmin >= view.quad_m + NM_QUAD_EXT || (ix->d_quad_small && ix->quad_small_m && kmin >= ix->quad_small_m + NM_QUAD_EXT);
}

// Expected table lines per position when the sites read the table with cores of m bases (windows of w = m + 4): one line
// per group of G = kmin - w + 5 positions, plus what the positions cost that no window settles.  f = share of repeated
// windows among the w-mers of a text of n symbols (uniform model); the first and last position of a group lie in one
// window, the others in at least two.
static double nm_site_cost(uint64_t n, uint32_t m, uint32_t kmin, uint32_t d_cap, double next_cost) {
    const uint32_t w = m + NM_QUAD_EXT;
    uint32_t d = kmin - w;
    if (d > d_cap) d = d_cap;
    const double G = d + 5.0;
    const double f = 1.0 - exp(-(double)n / pow(4.0, (double)w));
    // d = 0: positions 0, 1, 3, 4 of a group lie in one window each, position 2 in none
    const double open = d == 0 ? (4.0 * f + 1.0) / 5.0 : (2.0 * f + (G - 2.0) * f * f) / G;
    return 1.0 / G + open * next_cost;
}

// which quad table the sites of this launch read (view.quad) and which one backs them up in k_resolve (view.quad2)
static void nm_pick_site_tables(nm_index *ix, nm_view &view, uint32_t kmin) {
    const uint64_t *big = view.quad, *small = (const uint64_t *)ix->d_quad_small;
    const uint32_t big_m = view.quad_m, small_m = ix->quad_small_m;
    view.quad2 = nullptr;
    view.quad2_m = 0;
    const bool big_ok = big && kmin >= big_m + NM_QUAD_EXT, small_ok = small && small_m && kmin >= small_m + NM_QUAD_EXT;
    if (!small_ok) return;
    bool use_small;
    if (ix->site_table == 1 && big_ok) use_small = false;
    else if (ix->site_table == 2 || !big_ok) use_small = true;
    else {
        const double walk = 4.0;                                              // seed entry + rank lines of a short walk
        const double f_big = 1.0 - exp(-(double)ix->h.n / pow(4.0, (double)(big_m + NM_QUAD_EXT)));
        use_small = nm_site_cost(ix->h.n, small_m, kmin, ix->site_d_cap, 1.0 + f_big * walk) < nm_site_cost(ix->h.n, big_m, kmin, ix->site_d_cap, walk);
    }
    if (use_small) {
        view.quad = small; view.quad_m = small_m;
        if (big_ok) { view.quad2 = big; view.quad2_m = big_m; }
    }
}

// k_sites -> repeat probes where the bitmap is dense -> k_resolve, over positions [0, n).  Range mode: kmin .. kmax.
// List mode (d_list != nullptr): kmin = the first listed length, kmax = the longest.
template <bool BIG>
static int launch_sites(nm_index *ix, const nm_view &view_in, const void *d_seq, uint64_t seq_len, uint64_t n, uint32_t kmin, uint32_t kmax,
                        void *d_out, int elem_bytes, uint64_t *d_status, hipStream_t st, bool status_ready,
                        const uint32_t *d_list = nullptr, uint32_t n_list = 0, bool hash = true) {
    int rc = nm_prepare_enc(ix, seq_len);
    if (rc != NM_OK) return rc;
    const nm_enc_word *enc = (const nm_enc_word *)ix->cur->enc.p;
    const uint64_t n_need = (n + 63) / 64;
    if ((rc = nm_grow(ix->cur->need, (n_need + 1) * sizeof(uint64_t))) != NM_OK) return rc;
    uint64_t *need = (uint64_t *)ix->cur->need.p;
    unsigned long long *work = (unsigned long long *)ix->cur->work.p;
    nm_view view = view_in;
    nm_pick_site_tables(ix, view, kmin);
    ix->last_site_m = view.quad_m;
    uint32_t d = kmin - (view.quad_m + NM_QUAD_EXT);
    if (d > ix->site_d_cap) d = ix->site_d_cap;
    const uint32_t bp = nm_site_block_positions(d);
    const dim3 sgrid((unsigned)((n + bp - 1) / bp)), sblock(NM_SITE_BLOCK);
    size_t lds = nm_site_lds_bytes(d, kmax);
    // NEWMAP_AMD_SITES_BLOCKS_PER_CU (measurement knob): cap the blocks of k_sites a CU holds by asking for more LDS than it
    // needs -- the lookups reach their line rate with 16 waves per CU (tools/gather_ceiling), and wave slots left free let the
    // latency-bound kernels of the neighbouring streams (probes, k_resolve) start at once on repeat-rich input
    if (ix->sites_blocks_per_cu > 0) {
        const size_t per_block = (size_t)(160u << 10) / (size_t)ix->sites_blocks_per_cu;
        const size_t want = per_block > 1024 ? (per_block - 512) & ~(size_t)255 : lds;
        if (want > lds && want <= (64u << 10)) lds = want;
    }
    ix->last_kernel = 5;
    if ((rc = nm_grow(ix->cur->hashp, (uint64_t)sgrid.x * sizeof(uint64_t))) != NM_OK) return rc;
    uint64_t *hash_part = nullptr;                             // set below when k_sites fingerprints the segment itself
    // Input that has shown long repeats before (the latch the fine probes set): the probes are walks of up to kmax + 511
    // dependent steps -- bound by latency, not by lines -- so they run on a second stream BESIDE k_sites (every stride:
    // the bitmap that would gate them is not there yet) and k_resolve waits for both.  Otherwise they follow k_sites
    // and look only at the strides it left mostly open -- on input without long repeats that is none at all.
    const uint32_t *probe = nullptr;
    const bool repeats_met = ix->h_repeats_seen && *(volatile uint32_t *)ix->h_repeats_seen != 0;
    const bool beside = ix->repeat_probes && repeats_met && ix->cur->side && ix->probes_beside && n >= (1u << 16);
    nm_enc_word *enc_out = (nm_enc_word *)ix->cur->enc.p;      // k_sites leaves the encoded words for the probes and k_resolve
    if (beside) {
        // (the probes start before k_sites has encoded anything: this launch takes the separate encode pass)
        if ((rc = nm_encode(ix, d_seq, seq_len, st, status_ready ? nullptr : d_status)) != NM_OK) return rc;
        if (hash && (rc = nm_hash_positions(ix, n, d_status, st)) != NM_OK) return rc;
        enc_out = nullptr;
        hash = false;
        HIP_TRY(hipEventRecord(ix->cur->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(ix->cur->side, ix->cur->ev_fork, 0));
        if ((rc = nm_launch_probes<BIG>(ix, view, n, kmax, ix->cur->side, &probe, nullptr)) != NM_OK) return rc;
        HIP_TRY(hipEventRecord(ix->cur->ev_join, ix->cur->side));
    } else if (!status_ready && (rc = nm_reset_status(ix, d_status, st)) != NM_OK) return rc;
    if (hash) hash_part = (uint64_t *)ix->cur->hashp.p;
    {
        nm_timed timed(ix, st);
#define NM_LAUNCH_SITES(STATS_, LIST_) hipLaunchKernelGGL((k_sites<BIG, STATS_, LIST_>), sgrid, sblock, lds, st, view, (const uint8_t *)d_seq, seq_len, \
                                                          enc_out, ix->cur->enc_words, n, kmin, kmax, d, d_out, elem_bytes, d_status, need, work, d_list, n_list, hash_part)
        if (d_list) { if (ix->count_steps) NM_LAUNCH_SITES(true, true); else NM_LAUNCH_SITES(false, true); }
        else        { if (ix->count_steps) NM_LAUNCH_SITES(true, false); else NM_LAUNCH_SITES(false, false); }
#undef NM_LAUNCH_SITES
    }
    if (beside) HIP_TRY(hipStreamWaitEvent(st, ix->cur->ev_join, 0));
    else if (ix->repeat_probes && (rc = nm_launch_probes<BIG>(ix, view, n, kmax, st, &probe, need)) != NM_OK) return rc;
    const dim3 rgrid((unsigned)((n_need + NM_RES_WORDS - 1) / NM_RES_WORDS)), rblock(NM_RES_BLOCK);
#define NM_LAUNCH_RES(STATS_, LIST_) hipLaunchKernelGGL((k_resolve<BIG, STATS_, LIST_>), rgrid, rblock, 0, st, view, enc, n, kmin, kmax, d_out, elem_bytes, \
                                                        d_status, (const uint64_t *)need, n_need, probe, (const unsigned long long *)work, seq_len, d_list, n_list, (const uint64_t *)hash_part, (uint32_t)sgrid.x)
    {
        nm_timed timed(ix, st, 4);
        if (d_list) { if (ix->count_steps) NM_LAUNCH_RES(true, true); else NM_LAUNCH_RES(false, true); }
        else        { if (ix->count_steps) NM_LAUNCH_RES(true, false); else NM_LAUNCH_RES(false, false); }
    }
#undef NM_LAUNCH_RES
    return NM_OK;
}

// range mode over positions [0, num_kmers) of a segment: the sites, or (--norc, kmin outside the tables' windows, A/B)
// the encode pass + one lane per position.  status_ready: the caller has reset the status words already.
template <bool BIG, bool RC>
static int launch_min_unique(nm_index *ix, const nm_view &view, const void *d_seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmin,
                             uint32_t kmax, void *d_out, int elem_bytes, uint64_t *d_status, hipStream_t st, bool status_ready, bool hash = true) {
    if (RC && nm_sites_apply(ix, view, kmin))
        return launch_sites<BIG>(ix, view, d_seq, seq_len, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st, status_ready, nullptr, 0, hash);
    int rc = nm_encode(ix, d_seq, seq_len, st, status_ready ? nullptr : d_status);
    if (rc != NM_OK) return rc;
    if (hash && (rc = nm_hash_positions(ix, num_kmers, d_status, st)) != NM_OK) return rc;
    const dim3 block(NM_BLOCK);
    const nm_enc_word *enc = (const nm_enc_word *)ix->cur->enc.p;
    // one lane per position; on both strands the repeat probes run first (every stride: there is no bitmap to gate them)
    const uint32_t *settled = nullptr;
    if (RC && ix->repeat_probes) {
        rc = nm_launch_probes<BIG>(ix, view, num_kmers, kmax, st, &settled);
        if (rc != NM_OK) return rc;
    }
    nm_timed timed(ix, st);
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
    if ((rc = nm_lane_for(ix, st)) != NM_OK) return rc;
    if (num_kmers == 0) return nm_reset_status(ix, d_status, st);
    nm_view view;
    if ((rc = nm_view_for(ix, kmin, &view)) != NM_OK) return rc;
    nm_timed whole(ix, st, 1);
    if (ix->big) rc = use_revcomp ? launch_min_unique<true, true>(ix, view, d_seq, seq_len, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st, false)
                                  : launch_min_unique<true, false>(ix, view, d_seq, seq_len, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st, false);
    else         rc = use_revcomp ? launch_min_unique<false, true>(ix, view, d_seq, seq_len, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st, false)
                                  : launch_min_unique<false, false>(ix, view, d_seq, seq_len, num_kmers, kmin, kmax, d_out, elem_bytes, d_status, st, false);
    if (rc != NM_OK) return rc;
    HIP_TRY(hipGetLastError());
    return nm_lane_done(ix, st);
}

template <bool BIG, bool RC>
static void launch_fixed_k(nm_index *ix, const nm_view &view, uint64_t seq_len, uint64_t first, uint64_t num_kmers, const uint32_t *d_ks, uint32_t nk,
                           void *d_out, int elem_bytes, uint64_t *d_status, hipStream_t st) {
    const dim3 grid(nm_grid(num_kmers - first)), block(NM_BLOCK);      // positions [first, num_kmers)
    const nm_enc_word *enc = (const nm_enc_word *)ix->cur->enc.p;
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
    if ((rc = nm_lane_for(ix, st)) != NM_OK) return rc;
    if ((rc = nm_reset_status(ix, d_status, st)) != NM_OK) return rc;
    if (num_kmers == 0) return NM_OK;
    nm_timed whole(ix, st, 1);
    if ((rc = nm_grow(ix->cur->ks, (uint64_t)nk * sizeof(uint32_t))) != NM_OK) return rc;
    HIP_TRY(hipMemcpyAsync(ix->cur->ks.p, ks, (uint64_t)nk * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    bool encoded = false;                                   // the range / sites launches below leave the segment's encoded words behind
    const uint32_t *d_ks = (const uint32_t *)ix->cur->ks.p;
    uint32_t kshort = ks[0];
    for (uint32_t i = 1; i < nk; i++) if (ks[i] < kshort) kshort = ks[i];
    nm_view view;
    if ((rc = nm_view_for(ix, kshort, &view)) != NM_OK) return rc;
    // ONE length K on both strands is range mode with kmin = kmax = K for every position whose K-mer lies inside
    // the data (same walk, same early stop at one occurrence, same ambiguity rule): those positions take the
    // range kernels with their tables and repeat probes; the up to K-1 positions at the end of the data, whose
    // k-mer the reference truncates (search.py:590), keep the list kernel.
    uint64_t first = 0;
    if (nk == 1 && use_revcomp && ix->list_via_range && view.quad && seq_len >= ks[0]) {
        const uint64_t head = num_kmers < seq_len - ks[0] + 1 ? num_kmers : seq_len - ks[0] + 1;
        if (head) {
            rc = ix->big ? launch_min_unique<true, true>(ix, view, d_seq, seq_len, head, ks[0], ks[0], d_out, elem_bytes, d_status, st, true, false)
                         : launch_min_unique<false, true>(ix, view, d_seq, seq_len, head, ks[0], ks[0], d_out, elem_bytes, d_status, st, true, false);
            if (rc != NM_OK) return rc;
            first = head;
            encoded = true;
        }
    }
    // several lengths, the first one at least as long as a quad table's window: the sites with the FIRST length in the
    // place of kmin (a position whose first-length k-mer contains a window that occurs once is unique at that length:
    // the answer, whatever the other lengths are), the list form of k_resolve for the rest
    if (nk > 1 && use_revcomp && ix->list_via_range && nm_sites_apply(ix, view, ks[0]) && seq_len >= kmax) {
        const uint64_t head = num_kmers < seq_len - kmax + 1 ? num_kmers : seq_len - kmax + 1;
        if (head) {
            rc = ix->big ? launch_sites<true>(ix, view, d_seq, seq_len, head, ks[0], kmax, d_out, elem_bytes, d_status, st, true, d_ks, nk, false)
                         : launch_sites<false>(ix, view, d_seq, seq_len, head, ks[0], kmax, d_out, elem_bytes, d_status, st, true, d_ks, nk, false);
            if (rc != NM_OK) return rc;
            first = head;
            encoded = true;
        }
    }
    if (first < num_kmers) {
        if (!encoded && (rc = nm_encode(ix, d_seq, seq_len, st)) != NM_OK) return rc;
        if (ix->big) { if (use_revcomp) launch_fixed_k<true, true>(ix, view, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status, st); else launch_fixed_k<true, false>(ix, view, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status, st); }
        else         { if (use_revcomp) launch_fixed_k<false, true>(ix, view, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status, st); else launch_fixed_k<false, false>(ix, view, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status, st); }
    }
    if ((rc = nm_hash_positions(ix, num_kmers, d_status, st)) != NM_OK) return rc;   // (list mode: one pass over the encoded words)
    HIP_TRY(hipGetLastError());
    return nm_lane_done(ix, st);
}

// host-buffer wrappers ---------------------------------------------------------------------

static int nm_finish_segment(nm_index *ix, void *out, uint64_t out_bytes, uint64_t *n_ambiguous, uint64_t *bad_pos) {
    uint64_t status[NM_STATUS_WORDS];
    HIP_TRY(hipMemcpyAsync(status, ix->status.p, sizeof status, hipMemcpyDeviceToHost, ix->stream));
    if (out_bytes) HIP_TRY(hipMemcpyAsync(out, ix->out.p, out_bytes, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    if (n_ambiguous) *n_ambiguous = status[0];
    if (bad_pos) *bad_pos = status[2];
    ix->last_fingerprint = status[NM_STATUS_HASH];
    if (status[1]) {
        nm_set_error("a generated k-mer was not found in the index (first at segment position %llu); "
                     "possibly a mismatch between the sequence and the index", (unsigned long long)status[2]);
        return NM_E_KMER_NOT_FOUND;
    }
    return NM_OK;
}

extern "C" int nm_guard_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks, uint32_t nk,
                                    int range_mode, uint32_t initial_len, int use_revcomp, uint64_t *d_status, void *stream);

// The host-buffer segment calls are the seam of newmap/search.py's binary_search / linear_search, which raise on an absent
// probe (:699-722): unless the segment is, by length and fingerprint, a whole indexed record, the staged segment goes
// through the exact guard as well.  NM_OPT_SEGMENT_GUARD = 0: the caller checks whole records itself (the drivers).
static int nm_seam_guard(nm_index *ix, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks, uint32_t nk, int range_mode,
                         uint32_t initial_len, int use_revcomp, uint64_t *bad_pos) {
    if (!ix->segment_guard || num_kmers == 0) return NM_OK;
    if (num_kmers == seq_len && nm_index_has_record(ix, seq_len, ix->last_fingerprint)) return NM_OK;
    const uint64_t fp = ix->last_fingerprint;
    int rc = nm_guard_segment_dev(ix, ix->seq.p, seq_len, num_kmers, ks, nk, range_mode, initial_len, use_revcomp, (uint64_t *)ix->status.p, ix->stream);
    if (rc == NM_OK) rc = nm_finish_segment(ix, nullptr, 0, nullptr, bad_pos);
    ix->last_fingerprint = fp;
    return rc;
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
    if ((rc = nm_finish_segment(ix, out, out_bytes, n_ambiguous, bad_pos)) != NM_OK) return rc;
    const uint32_t two[2] = {kmin, kmax};
    return nm_seam_guard(ix, seq_len, num_kmers, two, 2, 1, initial_len, use_revcomp, bad_pos);
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
    if ((rc = nm_finish_segment(ix, out, out_bytes, n_ambiguous, bad_pos)) != NM_OK) return rc;
    return nm_seam_guard(ix, seq_len, num_kmers, ks, nk, 0, 0, use_revcomp, bad_pos);
}

// ---- the exact zero-count guard over one segment (records that are not among the indexed ones; include/newmap_amd.h) ----
extern "C" int nm_guard_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks, uint32_t nk,
                                    int range_mode, uint32_t initial_len, int use_revcomp, uint64_t *d_status, void *stream) {
    int rc = nm_check_segment_args(ix, seq_len, num_kmers, 4);
    if (rc != NM_OK) return rc;
    if (!ks || nk == 0 || (range_mode && nk != 2)) { nm_set_error("the guard takes kmin, kmax (range mode) or the list of lengths"); return NM_E_ARGUMENT; }
    for (uint32_t i = 0; i < nk; i++) if (ks[i] < 1) { nm_set_error("k-mer lengths must be >= 1"); return NM_E_ARGUMENT; }
    if (!d_status) { nm_set_error("d_status is required"); return NM_E_ARGUMENT; }
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t st = stream ? (hipStream_t)stream : ix->stream;
    if ((rc = nm_lane_for(ix, st)) != NM_OK) return rc;
    if ((rc = nm_reset_status(ix, d_status, st, false)) != NM_OK) return rc;   // (the counters of the search before it stay readable)
    if (num_kmers == 0) return NM_OK;
    ix->guard_segments++;
    if ((rc = nm_encode(ix, d_seq, seq_len, st)) != NM_OK) return rc;
    nm_view view = ix->view;                               // the walks start from the first base: no tables
    const uint32_t *d_ks = nullptr;
    uint32_t kmin = ks[0], kmax = ks[0];
    for (uint32_t i = 1; i < nk; i++) { if (ks[i] < kmin) kmin = ks[i]; if (ks[i] > kmax) kmax = ks[i]; }
    if (!range_mode) {
        if ((rc = nm_grow(ix->cur->ks, (uint64_t)nk * sizeof(uint32_t))) != NM_OK) return rc;
        HIP_TRY(hipMemcpyAsync(ix->cur->ks.p, ks, (uint64_t)nk * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        d_ks = (const uint32_t *)ix->cur->ks.p;
    }
    const dim3 grid(nm_grid(num_kmers)), block(NM_BLOCK);
    const nm_enc_word *enc = (const nm_enc_word *)ix->cur->enc.p;
    const uint32_t n_list = range_mode ? 0u : nk;
#define NM_LAUNCH_GUARD(BIG_, RC_) hipLaunchKernelGGL((k_guard<BIG_, RC_>), grid, block, 0, st, view, enc, seq_len, num_kmers, kmin, kmax, initial_len, d_ks, n_list, d_status)
    if (ix->big) { if (use_revcomp) NM_LAUNCH_GUARD(true, true); else NM_LAUNCH_GUARD(true, false); }
    else         { if (use_revcomp) NM_LAUNCH_GUARD(false, true); else NM_LAUNCH_GUARD(false, false); }
#undef NM_LAUNCH_GUARD
    HIP_TRY(hipGetLastError());
    return nm_lane_done(ix, st);
}

extern "C" int nm_guard_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks, uint32_t nk,
                                int range_mode, uint32_t initial_len, int use_revcomp, uint64_t *bad_pos) {
    int rc = nm_check_segment_args(ix, seq_len, num_kmers, 4);
    if (rc != NM_OK) return rc;
    if (!seq && seq_len) { nm_set_error("null buffer"); return NM_E_ARGUMENT; }
    if ((rc = nm_stage_segment(ix, seq, seq_len, 0)) != NM_OK) return rc;
    rc = nm_guard_segment_dev(ix, ix->seq.p, seq_len, num_kmers, ks, nk, range_mode, initial_len, use_revcomp, (uint64_t *)ix->status.p, ix->stream);
    if (rc != NM_OK) return rc;
    return nm_finish_segment(ix, nullptr, 0, nullptr, bad_pos);
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
    if ((rc = nm_lane_for(ix, ix->stream)) != NM_OK) return rc;
    if ((rc = nm_encode(ix, ix->seq.p, seq_len, ix->stream)) != NM_OK) return rc;
    hipLaunchKernelGGL(k_upper, dim3(nm_grid(num_kmers)), dim3(NM_BLOCK), 0, ix->stream, (const nm_enc_word *)ix->cur->enc.p,
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
    if ((rc = nm_grow(ix0->out, out_bytes + 64)) != NM_OK || (rc = nm_grow(ix0->lanes[0].ks, (uint64_t)nk * 4)) != NM_OK) { cleanup(); return rc; }
    if ((rc = nm_reset_status(ix0, (uint64_t *)ix0->status.p, st)) != NM_OK) { cleanup(); return rc; }
    if (hipMemcpyAsync(ix0->lanes[0].ks.p, ks, (uint64_t)nk * 4, hipMemcpyHostToDevice, st) != hipSuccess) { cleanup(); nm_set_error("copy to device failed"); return NM_E_DEVICE; }
    if (num_kmers) {
        const uint32_t list_n = range_mode ? 0u : nk;
        if (use_revcomp) hipLaunchKernelGGL(k_multi<true>, dim3(nm_grid(num_kmers)), dim3(NM_BLOCK), 0, st, a, seq_len, num_kmers, kmin, kmax, (const uint32_t *)ix0->lanes[0].ks.p, list_n, ix0->out.p, elem_bytes, (uint64_t *)ix0->status.p);
        else             hipLaunchKernelGGL(k_multi<false>, dim3(nm_grid(num_kmers)), dim3(NM_BLOCK), 0, st, a, seq_len, num_kmers, kmin, kmax, (const uint32_t *)ix0->lanes[0].ks.p, list_n, ix0->out.p, elem_bytes, (uint64_t *)ix0->status.p);
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

// record fingerprints (nm_hash.h) ---------------------------------------------------------------

extern "C" int nm_index_has_record(const nm_index *ix, uint64_t length, uint64_t hash) {
    if (!ix) return 0;
    const nm_record_entry key{length, hash};
    return std::binary_search(ix->records.begin(), ix->records.end(), key, [](const nm_record_entry &a, const nm_record_entry &b) {
        return a.length != b.length ? a.length < b.length : a.hash < b.hash; }) ? 1 : 0;
}

extern "C" uint64_t nm_index_records(const nm_index *ix, uint64_t *lengths, uint64_t *hashes, uint64_t capacity) {
    if (!ix) return 0;
    for (uint64_t i = 0; i < ix->records.size() && i < capacity; i++) {
        if (lengths) lengths[i] = ix->records[i].length;
        if (hashes) hashes[i] = ix->records[i].hash;
    }
    return ix->records.size();
}

extern "C" uint64_t nm_fingerprint_join(uint64_t ha, uint64_t len_a, uint64_t hb) {
    return nm_hash_join(ha, len_a / 64, hb);               // (len_a must be a multiple of 64: a segment starts at a word of its record)
}

extern "C" uint64_t nm_fingerprint_sequence(const uint8_t *seq, uint64_t len) {
    uint64_t h = 0, pw = 1;
    for (uint64_t w = 0; w * 64 < len; w++) {
        uint64_t lo = 0, hi = 0, amb = 0;
        for (uint64_t j = 0; j < 64 && w * 64 + j < len; j++) {
            const uint32_t u = seq[w * 64 + j] & 0xDFu;
            const uint32_t c = u == 'A' ? 0 : (u == 'C' ? 1 : (u == 'G' ? 2 : (u == 'T' ? 3 : 4)));
            if (c > 3) amb |= 1ULL << j;
            else { lo |= (uint64_t)(c & 1u) << j; hi |= (uint64_t)(c >> 1) << j; }
        }
        h += nm_hash_word(lo, hi, amb) * pw;
        pw *= NM_HASH_R;
    }
    return h;
}

// nm_kernels.hip.h -- the device side of the engine: every kernel and the device helpers they share (part of nm_engine.hip;
// the per-position logic itself lives in nm_core.h).
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
// Measurement builds (make measure: -DNM_MEASURE, libnewmap_amd_measure.so, loaded by tools/ only) keep two switches of
// NM_OPT_SEED_POLICY that cut work out of the kernels -- and give WRONG results: 0x100 no walks in k_resolve, 0x200 no table
// load in k_sites.  The product library has no such code and rejects the bits.
#ifdef NM_MEASURE
#define NM_CUT(policy, bit) (((policy) & (bit)) != 0)
#else
#define NM_CUT(policy, bit) false
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

#define NM_WORK_WORDS 16            /* handle-owned counters: [1..4] probe tally, [5] NM_WORK_OPEN, [6..] the sweep's lists */
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
        go[s] = base + (uint64_t)g * G < num_kmers && nm_site_core_valid(win[s], m) && !NM_CUT(ix.seed_policy, 0x200u);
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
    nm_tally t = {0, 0, 0, 0};
    bool any_err = false;
    uint64_t err_pos = ~0ULL;
    const bool use_dict = !LIST && ix.dict != nullptr && kmin >= ix.dict_len && !(ix.seed_policy & 0x1000u);
    bool dict_done = false;
    // ---- phase 3: second chance.  A position no site settled asks the table with the longer cores (nm_second_chance; its
    // window is in LDS).  All lookups of the block are in flight together; a block with many open positions sits in a
    // long repeat and skips this.  With a dictionary: first this (one line settles most), then the dictionary for what is
    // left (seed_policy bit 0x2000, measurement knob: the dictionary alone).
    if (!(use_dict && (ix.seed_policy & 0x2000u)) && ix.quad2 != nullptr && kmin >= ix.quad2_m + NM_QUAD_EXT && s_open_total && s_open_total <= NM_SITE_CHANCE_MAX) {
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
    // ---- phase 3D (range mode with a dictionary, nm_core.h): an open position looks its x-mer up in the repeat dictionary --
    // ONE 128-byte bucket, read by eight neighbouring lanes (16 bytes each, one line per load instruction and position).
    // A miss: the x-mer occurs once, the element is kmin.  A hit: the position walks from an interval x bases deep
    // (phase 4D).  Up to four positions per lane group are in flight.
    if (use_dict && s_open_total && s_open_total <= NM_SITE_CHANCE_MAX) {
        __shared__ uint32_t s_wpos[NM_SITE_CHANCE_MAX];
        __shared__ uint64_t s_went[NM_SITE_CHANCE_MAX];
        __shared__ uint32_t s_wn;
        gather_open();
        const uint32_t n_q = s_qn, x = ix.dict_len;
        __shared__ uint64_t s_dkey[NM_SITE_CHANCE_MAX];
        __shared__ uint32_t s_dbkt[NM_SITE_CHANCE_MAX];
        if (tid == 0) s_wn = 0;
        if (tid < n_q) {                                               // one lane per open position: its key and bucket
            const uint64_t key = nm_dict_key(lds_window(s_q[tid]), x);
            s_dkey[tid] = key;
            s_dbkt[tid] = (uint32_t)nm_dict_bucket(key, ix.dict_bits);
        }
        __syncthreads();
        const uint32_t sub = tid & 7u, gbase = (tid & 63u) & ~7u;
        for (uint32_t r0 = 0; r0 < n_q; r0 += 4 * (NM_SITE_BLOCK / 8)) {  // eight lanes per position read its bucket; four positions per group in flight
            uint64_t key[4], kk[4], vv[4];
            bool have[4];
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                const uint32_t qi = r0 + r * (NM_SITE_BLOCK / 8) + (tid >> 3);
                have[r] = qi < n_q;
                kk[r] = NM_DICT_EMPTY; vv[r] = 0; key[r] = 0;
                if (have[r]) {
                    key[r] = s_dkey[qi];
                    const nm_u64x2 v = nm_quad_load16((uint64_t)(ix.dict + ((uint64_t)s_dbkt[qi] * NM_DICT_SLOTS + sub) * 2));
                    kk[r] = v.x; vv[r] = v.y;
                    if (sub == 0) n_entries += 2;                 // counted as ONE slot (key + entry, 16 B): what a lookup needs; the bucket's other seven are the layout's cost
                }
            }
#pragma unroll
            for (uint32_t r = 0; r < 4; r++) {
                if (r0 + r * (NM_SITE_BLOCK / 8) >= n_q) break;               // (uniform: no position left for this turn)
                const uint32_t m8 = (uint32_t)((__ballot(have[r] && kk[r] == key[r]) >> gbase) & 0xFFu);
                const uint32_t e8 = (uint32_t)((__ballot(have[r] && kk[r] == NM_DICT_EMPTY) >> gbase) & 0xFFu);
                const int src = m8 ? (int)(gbase + (uint32_t)__builtin_ctz(m8)) : (int)gbase;
                uint64_t entry = __shfl(vv[r], src, NM_WAVE);
                if (have[r] && sub == 0) {
                    const uint32_t rel = s_q[r0 + r * (NM_SITE_BLOCK / 8) + (tid >> 3)];
                    bool hit = m8 != 0;
                    if (!hit && !e8) hit = nm_dict_find(ix, key[r], entry);       // (a full bucket without the key: it may have spilled over)
                    if (hit) {
                        const uint32_t slot = atomicAdd(&s_wn, 1u);
                        s_wpos[slot] = rel;
                        s_went[slot] = entry;
                    } else {
                        nm_store(out, elem_bytes, base + rel, kmin);
                        atomicAnd(&s_need[rel >> 5], ~(1u << (rel & 31)));
                        atomicSub(&s_open_total, 1u);
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 4D: the positions whose x-mer is repeated walk on from its interval (a few: here; many: a long repeat, k_resolve)
        const uint32_t n_w = s_wn;
        if (n_w <= NM_SITE_WALK_MAX && kmax <= NM_SITE_LA_MAX && !(ix.seed_policy & 0x800u) && !NM_CUT(ix.seed_policy, 0x100u)) {
            if (tid < n_w) {
                const uint32_t relw = s_wpos[tid];
                uint64_t lo, hi;
                bool amb0 = false, err = false;
                uint32_t v;
                if (nm_seed_decode(s_went[tid], lo, hi)) v = nm_min_unique_walk<BIG, true>(ix, s_enc, relw, lds_window(relw), 0, lo, hi, x, kmin, kmax, err, t);
                else v = nm_min_unique_one<BIG, true>(ix, s_enc, relw, kmin, kmax, amb0, err, t);
                if (err) { any_err = true; err_pos = base + relw; }
                nm_store(out, elem_bytes, base + relw, v);
                atomicAnd(&s_need[relw >> 5], ~(1u << (relw & 31)));
            }
            if (tid == 0) s_open_total = 0;
            dict_done = true;
        }
        if (tid == 0) s_qn = 0;
        __syncthreads();
    }
    // ---- phase 4: a few open positions (the rule outside long repeats): the block finishes them itself -- seed table +
    // walk -- and hands an empty bitmap on.  Many: they stay for the repeat probes and k_resolve.
    const uint32_t open_total = s_open_total;
    // (the walks read the block's staged words -- positions relative to its first base -- so the lookahead of the
    // longest walk must have been staged: kmax <= NM_SITE_LA_MAX)
    // (seed_policy bit 0x800, measurement knob: the blocks never walk themselves, every open position goes to k_resolve)
    const bool self = !dict_done && open_total && open_total <= NM_SITE_WALK_MAX && kmax <= NM_SITE_LA_MAX && !(ix.seed_policy & 0x800u) && !NM_CUT(ix.seed_policy, 0x100u);
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
        if (base + 64ull * i < num_kmers) need[w0 + i] = (self || dict_done) ? 0ULL : ((uint64_t)s_need[2 * i] | ((uint64_t)s_need[2 * i + 1] << 32));

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

// ---- which kernel finishes the open positions of a launch: decided on the device, per word and per launch -------------------
// k_open_words sorts the words of the need bitmap by their number of open positions.  Dense words (repeat family members:
// stretches of open positions whose least unique strings end at common points) go to the sweep, which shares the walks
// (k_sweep); in sparse words (fewer than 24 open positions: the scattered repeated windows of ordinary sequence, what the repeat
// probes leave at the ends of tandem arrays) every position walks for itself, two bases per step (k_resolve on the bitmap of
// those words).  A launch whose open positions lie mostly in sparse words is k_resolve's altogether: the few dense words it
// has are long chains (the last kmax positions of a tandem array: one walk of up to kmax steps, then a step per position) that
// would keep the launch waiting for their latency.  Both kernels are launched; what has nothing to do returns at once.
#define NM_WORK_TAKEN 6             /* entries the waves of k_sweep have taken */
#define NM_WORK_LIST 8              /* [8 + c]: number of words of class c that k_open_words listed, [12 + c]: their open positions */
#define NM_SWEEP_CLASSES 4u         /* words by number of open positions: >= 48, >= 32, >= 24 (the long chains start first), sparse or scattered */
__device__ __forceinline__ bool nm_sweep_mode(const unsigned long long *work) {
    return work[NM_WORK_LIST + 4] + work[NM_WORK_LIST + 5] + work[NM_WORK_LIST + 6] >= work[NM_WORK_LIST + 7];  // most open positions lie in dense words
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
#define NM_RES_HASH_BLOCKS 32u
template <bool BIG, bool STATS, bool LIST>
__global__ __launch_bounds__(NM_RES_BLOCK) void k_resolve(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t num_kmers,
                                                          uint32_t kmin, uint32_t kmax, void *__restrict__ out, int elem_bytes,
                                                          uint64_t *__restrict__ status, const uint64_t *__restrict__ need,
                                                          uint64_t n_need, const uint32_t *__restrict__ probe,
                                                          const unsigned long long *__restrict__ work,
                                                          uint64_t seq_len, const uint32_t *__restrict__ list, uint32_t n_list,
                                                          const uint64_t *__restrict__ hash_part, uint32_t n_hash_part, int after_sweep,
                                                          volatile uint32_t *open_seen, uint32_t *__restrict__ seen_latch, const uint64_t *__restrict__ need_sparse) {
    // (after_sweep: k_sweep ran before this kernel, added the fingerprint up, and took the launch if its words are dense)
    if (after_sweep) {
        hash_part = nullptr;
        if (work[NM_WORK_OPEN] == 0) return;
        if (nm_sweep_mode(work)) need = need_sparse;       // the dense words were k_sweep's
    }
    // the segment's fingerprint: the partial sums of k_sites' blocks (nm_hash.h), added up by the first blocks of this grid -- a
    // slice each, so that no lane reads more than a few values one after the other (one block reading them all was a chain of
    // ~100 dependent loads: 20 - 50 us per launch), and at most NM_RES_HASH_BLOCKS atomics meet on the status word
    const uint32_t hb = gridDim.x < NM_RES_HASH_BLOCKS ? gridDim.x : NM_RES_HASH_BLOCKS;
    if (hash_part && blockIdx.x < hb) {
        uint64_t term = 0;
        for (uint32_t i = blockIdx.x * NM_RES_BLOCK + threadIdx.x; i < n_hash_part; i += hb * NM_RES_BLOCK) term += hash_part[i];
        for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, NM_WAVE);
        __shared__ uint64_t s_term[NM_RES_BLOCK / 64];
        if ((threadIdx.x & 63) == 0) s_term[threadIdx.x >> 6] = term;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t sum = 0;
            for (uint32_t i = 0; i < NM_RES_BLOCK / 64; i++) sum += s_term[i];
            if (sum) atomicAdd((unsigned long long *)&status[NM_STATUS_HASH], (unsigned long long)sum);
        }
    }
    if (work[NM_WORK_OPEN] == 0) return;                   // every block of k_sites finished its own positions
    // tell the host (pinned word [1], once per handle, as the fine probes do for long repeats) that this input leaves positions
    // open: from then on a launch also lists the open words and offers them to k_sweep.  Input that never does -- a genome
    // without repeats: every block of k_sites finishes its own few -- is spared the two launches.
    if (open_seen && blockIdx.x == 0 && threadIdx.x == 0 && seen_latch[1] == 0u) { seen_latch[1] = 1u; open_seen[1] = 1u; }
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
        const uint32_t n_walk = NM_CUT(ix.seed_policy, 0x100u) ? 0u : (q_n < NM_RES_QCAP ? q_n : NM_RES_QCAP);   // (measurement builds: no walks)
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

// ---- k_open_words + k_sweep: the positions k_sites left open, neighbours sharing their walks (nm_core.h "the sweep") -------
// k_open_words lists the words of the need bitmap that hold open positions, by class (above), and writes the bitmap of the
// sparse ones; k_sweep gives every lane one listed dense word at a time -- a word = one chain from its last open position down
// to its first -- the longest chains first, dealt to the waves in chunks (one atomic per chunk, the chunk staged in LDS), so
// that waves are dense whatever the open positions' distribution over the segment (repeats come in clusters: with a block per
// stretch of the bitmap the longest block set the launch's duration).  Every turn of the loop is ONE extension step for every
// lane that has work.  What the repeat probes decide is stored when a word is taken, as in k_resolve; a word's elements are
// collected in LDS and go back as one 64-byte line.
#define NM_SWEEP_CHUNK 128u         /* entries a wave takes at a time */
#define NM_SWEEP_BLOCK 256
#define NM_SWEEP_MAX_BLOCKS 2048u   /* 256 CUs x 8 blocks: the waves take words until the lists are used up */
#define NM_OPEN_PER_LANE 8u
#ifndef NM_SWEEP_WAVES
#define NM_SWEEP_WAVES 0            /* measurement builds: waves per SIMD the register allocation of k_sweep aims at (0 = as compiled) */
#endif
#if NM_SWEEP_WAVES
#define NM_SWEEP_ATTR __attribute__((amdgpu_waves_per_eu(NM_SWEEP_WAVES)))
#else
#define NM_SWEEP_ATTR
#endif
// a word is the sweep's when it has 24 open positions or more and at least half of them have an open right neighbour (a
// position is decided by one step only from the string its right neighbour left).  Fewer, or scattered, open positions -- the
// repeated windows of ordinary sequence, short runs each -- walk for themselves: their walks are a few steps long and
// k_resolve's single-interval steps are cheaper than the sweep's (measured on the uniform 3 Gbp genome with the small
// tables: the native driver 0.28 s with k_resolve, 0.33 s when words of 8 open positions went to the sweep).
__device__ __forceinline__ uint32_t nm_sweep_class(uint64_t bits) {
    const uint32_t n = nm_popc64(bits), chained = nm_popc64(bits & (bits >> 1));
    if (n < 24 || 2 * chained < n) return 3u;
    return n >= 48 ? 0u : (n >= 32 ? 1u : 2u);
}
// list[c * n_need + i] = i-th word of class c (in word order within a block: neighbouring words go to neighbouring lanes).
// The class is that of the positions the repeat probes leave: what lies inside a stretch repeated over more than kmax bases is
// 0 already, and between two probes that saw the same end every length is known (nm_probe_kstar) -- such a word is listed
// (its elements are stored when it is taken) but counts as sparse.
__global__ __launch_bounds__(NM_BLOCK) void k_open_words(const uint64_t *__restrict__ need, uint64_t n_need, const uint32_t *__restrict__ probe,
                                                         uint32_t *__restrict__ list, unsigned long long *__restrict__ work, uint64_t *__restrict__ need_sparse) {
    if (work[NM_WORK_OPEN] == 0) return;
    __shared__ uint32_t s_idx[NM_SWEEP_CLASSES][NM_BLOCK * NM_OPEN_PER_LANE];
    __shared__ uint32_t s_n[NM_SWEEP_CLASSES], s_base[NM_SWEEP_CLASSES], s_pos[NM_SWEEP_CLASSES];
    if (threadIdx.x < NM_SWEEP_CLASSES) { s_n[threadIdx.x] = 0; s_pos[threadIdx.x] = 0; }
    __syncthreads();
    const uint64_t first = (uint64_t)blockIdx.x * (NM_BLOCK * NM_OPEN_PER_LANE);
    for (uint32_t r = 0; r < NM_OPEN_PER_LANE; r++) {
        const uint64_t w = first + (uint64_t)r * NM_BLOCK + threadIdx.x;
        uint64_t bits = w < n_need ? need[w] : 0ULL;
        uint32_t cls = 3u;
        if (bits && probe) {
            const uint32_t wj = probe[w], wj1 = probe[w + 1], zeros = wj & 0xFFu;
            bits &= zeros >= 64 ? 0ULL : ~((1ULL << zeros) - 1ULL);
            if (!((wj1 >> 8) && (wj >> 8) == (wj1 >> 8) + NM_PROBE_STRIDE)) cls = nm_sweep_class(bits);
        } else if (bits) cls = nm_sweep_class(bits);
        if (w < n_need) need_sparse[w] = bits && cls == 3u ? need[w] : 0ULL;   // (as k_sites left it: k_resolve applies the probe words itself)
#pragma unroll
        for (uint32_t c = 0; c < NM_SWEEP_CLASSES; c++) {
            const bool mine = bits != 0 && cls == c;
            const uint64_t mask = __ballot(mine);
            if (!mask) continue;                           // (uniform)
            uint32_t at = 0;
            if ((threadIdx.x & 63) == (uint32_t)__builtin_ctzll(mask)) at = atomicAdd(&s_n[c], (uint32_t)__popcll(mask));
            at = __shfl(at, __builtin_ctzll(mask), NM_WAVE);
            if (mine) s_idx[c][at + (uint32_t)__popcll(mask & ((1ULL << (threadIdx.x & 63)) - 1ULL))] = (uint32_t)w;
            const uint32_t n_pos = wave_sum(mine ? nm_popc64(bits) : 0u);
            if ((threadIdx.x & 63) == 0) atomicAdd(&s_pos[c], n_pos);
        }
    }
    __syncthreads();
    if (threadIdx.x < NM_SWEEP_CLASSES && s_n[threadIdx.x]) {
        s_base[threadIdx.x] = (uint32_t)atomicAdd(&work[NM_WORK_LIST + threadIdx.x], (unsigned long long)s_n[threadIdx.x]);
        atomicAdd(&work[NM_WORK_LIST + NM_SWEEP_CLASSES + threadIdx.x], (unsigned long long)s_pos[threadIdx.x]);
    }
    __syncthreads();
    for (uint32_t c = 0; c < NM_SWEEP_CLASSES; c++)
        for (uint32_t i = threadIdx.x; i < s_n[c]; i += NM_BLOCK) list[(uint64_t)c * n_need + s_base[c] + i] = s_idx[c][i];
}

template <bool BIG, bool STATS, bool LIST>
__global__ __launch_bounds__(NM_SWEEP_BLOCK) NM_SWEEP_ATTR void k_sweep(nm_view ix, const nm_enc_word *__restrict__ enc, uint64_t num_kmers,
                                                          uint32_t kmin, uint32_t kmax, void *__restrict__ out, int elem_bytes,
                                                          uint64_t *__restrict__ status, const uint64_t *__restrict__ need,
                                                          uint64_t n_need, const uint32_t *__restrict__ open_list, const uint32_t *__restrict__ probe,
                                                          unsigned long long *__restrict__ work,
                                                          uint64_t seq_len, const uint32_t *__restrict__ list, uint32_t n_list,
                                                          const uint64_t *__restrict__ hash_part, uint32_t n_hash_part) {
    // the segment's fingerprint from the partial sums of k_sites' blocks, as in k_resolve
    const uint32_t hb = gridDim.x < NM_RES_HASH_BLOCKS ? gridDim.x : NM_RES_HASH_BLOCKS;
    if (hash_part && blockIdx.x < hb) {
        uint64_t term = 0;
        for (uint32_t i = blockIdx.x * NM_SWEEP_BLOCK + threadIdx.x; i < n_hash_part; i += hb * NM_SWEEP_BLOCK) term += hash_part[i];
        for (int off = 32; off > 0; off >>= 1) term += __shfl_down(term, off, NM_WAVE);
        __shared__ uint64_t s_term[NM_SWEEP_BLOCK / 64];
        if ((threadIdx.x & 63) == 0) s_term[threadIdx.x >> 6] = term;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t sum = 0;
            for (uint32_t i = 0; i < NM_SWEEP_BLOCK / 64; i++) sum += s_term[i];
            if (sum) atomicAdd((unsigned long long *)&status[NM_STATUS_HASH], (unsigned long long)sum);
        }
    }
    if (work[NM_WORK_OPEN] == 0 || !nm_sweep_mode(work)) return;   // every block of k_sites finished its own positions / sparse: k_resolve's
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    // the lists of the four classes, one after the other: entry e of the whole is entry e - first[c] of class c
    const uint32_t n0 = (uint32_t)work[NM_WORK_LIST], n1 = n0 + (uint32_t)work[NM_WORK_LIST + 1], n2 = n1 + (uint32_t)work[NM_WORK_LIST + 2];
    const uint32_t n_words = n2;                           // (the sparse words, class 3, are k_resolve's)
    // a wave's share of the list, staged in LDS when it is taken: the word, its open bits, its two probe words
    __shared__ uint32_t s_cur[NM_SWEEP_BLOCK / 64][NM_SWEEP_CHUNK], s_pj[NM_SWEEP_BLOCK / 64][NM_SWEEP_CHUNK], s_pj1[NM_SWEEP_BLOCK / 64][NM_SWEEP_CHUNK];
    __shared__ uint64_t s_bits[NM_SWEEP_BLOCK / 64][NM_SWEEP_CHUNK], s_wlo[NM_SWEEP_BLOCK / 64][NM_SWEEP_CHUNK], s_whi[NM_SWEEP_BLOCK / 64][NM_SWEEP_CHUNK];
    __shared__ uint64_t s_super[NM_MAX_SUPER * 4];         // C[c] + counts before each superblock: read every turn, by row
    if (BIG) {
        if (tid < ix.n_super * 4) s_super[tid] = ix.superC[tid];
        __syncthreads();
    }
    // The elements of a word are collected in LDS and written back as ONE 64-byte line when the word is done (uint8 output, the
    // usual case; else element by element).  A byte store per turn kept every turn waiting: the reads of a turn are waited for
    // with vmcnt(0), which on this hardware also counts the stores before them -- scattered single bytes, each a partial write
    // of a line -- and a turn then lasted ~20 us whatever it read.
    __shared__ __attribute__((aligned(16))) uint8_t s_out[NM_SWEEP_BLOCK][80];
    const bool can_buffer = elem_bytes == 1 && (((uintptr_t)out) & 15u) == 0;
    bool buffered = false;
    uint64_t out_word = 0;
    auto put = [&](uint64_t p_, uint32_t v_) {
        if (buffered && (p_ >> 6) == out_word) s_out[tid][p_ & 63] = (uint8_t)v_;
        else nm_store(out, elem_bytes, p_, v_);
    };
    uint32_t w_base = 0, w_next = 0, w_end = 0;            // (the same in every lane of the wave)
    nm_sweep_args args;
    args.kmin = kmin; args.kmax = kmax; args.seq_len = seq_len; args.list = LIST ? list : nullptr; args.n_list = n_list;
    args.sc = s_super;
    nm_sweep st;
    nm_sweep_begin(st, 0, 0, 0, 0);
    nm_tally t = {0, 0, 0, 0};
    bool any_err = false;
    uint64_t err_pos = ~0ULL;
    uint32_t turns_word = 0, turns_lane = 0, turns_wave = 0, turns_max = 0, words_taken = 0;    // counter build: shape of the chains
    for (;;) {
        // the lanes that have finished their word take the next words of the wave's share, in list order; a wave whose share
        // is used up takes the next NM_SWEEP_CHUNK entries of the list (one atomic per chunk: waves that start late, or
        // whose words were short, simply find less left) and stages them -- all its lanes read, once per chunk, what a
        // lane taking a word would otherwise wait for in the middle of everybody's chain
        const bool want = st.mode == NM_SW_IDLE && !st.bits;
        if (want && buffered) {                            // the word is done: its line goes back
            const nm_u64x2 *row = reinterpret_cast<const nm_u64x2 *>(&s_out[tid][0]);
            nm_u64x2 *dst = reinterpret_cast<nm_u64x2 *>((uint8_t *)out + out_word * 64);
            dst[0] = row[0]; dst[1] = row[1]; dst[2] = row[2]; dst[3] = row[3];
            buffered = false;
        }
        const uint64_t wmask = __ballot(want);
        if (wmask) {
            if (w_next == w_end) {
                const uint64_t act = __ballot(true);
                uint32_t base = 0;
                if (lane == (uint32_t)__builtin_ctzll(act)) base = (uint32_t)atomicAdd(&work[NM_WORK_TAKEN], (unsigned long long)NM_SWEEP_CHUNK);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                w_base = w_next = base < n_words ? base : n_words;
                w_end = base + NM_SWEEP_CHUNK < n_words ? base + NM_SWEEP_CHUNK : n_words;
                const uint32_t n_act = (uint32_t)__popcll(act), mine = (uint32_t)__popcll(act & ((1ULL << lane) - 1ULL));
                for (uint32_t i = mine; i < w_end - w_base; i += n_act) {
                    const uint32_t e = w_base + i;
                    const uint32_t cur = e < n0 ? open_list[e] : (e < n1 ? open_list[n_need + (e - n0)] : open_list[2 * n_need + (e - n1)]);
                    s_cur[wv][i] = cur;
                    s_bits[wv][i] = need[cur];
                    s_pj[wv][i] = probe ? probe[cur] : 0u;
                    s_pj1[wv][i] = probe ? probe[cur + 1] : 0u;
                    s_wlo[wv][i] = enc[cur].lo;
                    s_whi[wv][i] = enc[cur].hi;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (want) {
                if (w_next == w_end) break;                // the list is used up
                const uint32_t slot = w_next + (uint32_t)__popcll(wmask & ((1ULL << lane) - 1ULL));
                if (slot < w_end) {
                    const uint32_t i = slot - w_base;
                    const uint64_t cur = s_cur[wv][i];
                    uint64_t bits = s_bits[wv][i];
                    if (can_buffer && cur * 64 + 64 <= num_kmers) {        // what k_sites stored for the word's 64 positions
                        const nm_u64x2 *src = reinterpret_cast<const nm_u64x2 *>((const uint8_t *)out + cur * 64);
                        nm_u64x2 *row = reinterpret_cast<nm_u64x2 *>(&s_out[tid][0]);
                        row[0] = src[0]; row[1] = src[1]; row[2] = src[2]; row[3] = src[3];
                        buffered = true;
                        out_word = cur;
                    }
                    if (probe) {
                        const uint32_t wj = s_pj[wv][i], wj1 = s_pj1[wv][i];
                        const uint32_t zeros = wj & 0xFFu;         // positions repeated over more than kmax bases: element 0, as stored
                        bits &= zeros >= 64 ? 0ULL : ~((1ULL << zeros) - 1ULL);
                        const uint32_t kj = wj >> 8, kj1 = wj1 >> 8;
                        if (bits && kj1 && kj == kj1 + NM_PROBE_STRIDE) {      // the same end on both sides: every length is known (nm_probe_kstar)
                            for (; bits; bits &= bits - 1) {
                                const uint32_t o = (uint32_t)__builtin_ctzll(bits);
                                const uint32_t v = nm_sweep_element(enc, args, cur * 64 + o, kj - o);
                                if (v) put(cur * 64 + o, v);
                            }
                        }
                    }
                    nm_sweep_begin(st, cur, bits, s_wlo[wv][i], s_whi[wv][i]);
                    if (STATS) { if (turns_word > turns_max) turns_max = turns_word; turns_word = 0; words_taken++; }
                }
            }
            const uint32_t n_want = (uint32_t)__popcll(wmask);
            w_next = w_next + n_want < w_end ? w_next + n_want : w_end;
            if (st.mode == NM_SW_IDLE && !st.bits) continue;   // (nothing left in the share this turn, or a word the probes decide completely)
        }
        uint64_t p;
        uint32_t v;
        if (STATS) { turns_word++; turns_lane++; turns_wave++; }
        const uint32_t ret = nm_sweep_step<BIG>(ix, enc, args, st, p, v, t);
        if (ret & NM_SW_ERR) {
            bool amb0 = false, err = true;
            if (LIST) v = nm_fixed_k_one<BIG, true>(ix, enc, p, seq_len, list, n_list, amb0, err, t);   // (which listed k-mer is absent decides: the plain form)
            if (err) { any_err = true; if (p < err_pos) err_pos = p; }
        }
        if (ret & NM_SW_EMIT) put(p, v);
    }
    if (__ballot(any_err)) {
        if (any_err) atomicMin((unsigned long long *)&status[2], (unsigned long long)err_pos);
        if ((tid & 63) == 0) atomicOr((unsigned long long *)&status[1], 1ULL);
    }
    if (STATS) {
        const uint32_t a = wave_sum(t.steps), b = wave_sum(t.blocks), c = wave_sum(t.seeds);
        if (turns_word > turns_max) turns_max = turns_word;
        const uint32_t tl = wave_sum(turns_lane), nw = wave_sum(words_taken);
        uint32_t tw = turns_wave, tm = turns_max;          // (turns of the wave = of its lane that stayed longest)
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o1 = __shfl_down(tw, off, NM_WAVE), o2 = __shfl_down(tm, off, NM_WAVE); tw = o1 > tw ? o1 : tw; tm = o2 > tm ? o2 : tm; }
        if ((tid & 63) == 0 && (a | b | c)) {
            atomicAdd((unsigned long long *)&status[3], (unsigned long long)a);
            atomicAdd((unsigned long long *)&status[4], (unsigned long long)b);
            atomicAdd((unsigned long long *)&status[6], (unsigned long long)c);
            atomicAdd((unsigned long long *)&status[14], (unsigned long long)b);               // rank blocks (32 B) read HERE ([4]: by k_resolve too, 16-byte LF entries)
            atomicAdd((unsigned long long *)&status[15], (unsigned long long)c);               // seed entries read HERE ([6]: by k_resolve too)
            atomicAdd((unsigned long long *)&status[9], (unsigned long long)nw);               // words swept
            atomicAdd((unsigned long long *)&status[10], (unsigned long long)tl);              // turns, summed over the lanes
            atomicAdd((unsigned long long *)&status[11], (unsigned long long)tw * 64ull);      // turns of the waves x 64: [10] / [11] = share of busy lanes
            atomicMax((unsigned long long *)&status[12], (unsigned long long)tm);              // the longest chain of a word, in turns
            atomicMax((unsigned long long *)&status[13], (unsigned long long)tw);              // the longest wave, in turns
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

// ---- two-base LF blocks (nm_core.h: nm_lf2_interval), built at open in three passes -------------------------------------
// pass 1: one wave per 64 rows: every lane finds the dinucleotide in front of its row (two dependent reads: the row's
// symbol, then the symbol of the row it maps to), sixteen ballots make the indicator words; lane d writes entry d with its
// popcount in `base`.  pass 2 (k_lf2_chunk_sums / host scan) and pass 3 (k_lf2_finish) turn the counts into
// "constant + count before the block".
#define NM_LF2_CHUNK 1024u          /* blocks per chunk of the two-level prefix sum */
template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_lf2_bits(nm_view ix, nm_lf_entry *__restrict__ lf2, uint64_t first, uint64_t n_blocks) {
    const uint64_t b = first + ((blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x) >> 6);
    const uint32_t lane = threadIdx.x & 63;
    if (b >= n_blocks) return;
    const uint32_t d = nm_bwt2_code<BIG>(ix, b * 64 + lane);
    uint64_t mine = 0;
#pragma unroll
    for (uint32_t x = 0; x < 16; x++) {
        const uint64_t m = __ballot(d == x);
        if (lane == x) mine = m;
    }
    if (lane < 16) { nm_lf_entry e; e.base = nm_popc64(mine); e.bits = mine; lf2[b * 16 + lane] = e; }
}
// sums[chunk][d] = rows with dinucleotide d in the chunk's blocks
__global__ __launch_bounds__(NM_BLOCK) void k_lf2_chunk_sums(const nm_lf_entry *__restrict__ lf2, uint64_t n_blocks, uint64_t n_chunks, uint64_t *__restrict__ sums) {
    const uint64_t t = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;      // one lane per (chunk, d)
    if (t >= n_chunks * 16) return;
    const uint64_t chunk = t >> 4, d = t & 15;
    const uint64_t b1 = (chunk + 1) * NM_LF2_CHUNK < n_blocks ? (chunk + 1) * NM_LF2_CHUNK : n_blocks;
    uint64_t s = 0;
    for (uint64_t b = chunk * NM_LF2_CHUNK; b < b1; b++) s += lf2[b * 16 + d].base;
    sums[t] = s;
}
// starts[chunk][d] = rows with d before the chunk (exclusive scan of the sums, done on the host: n_chunks is small);
// base = C[c2] + rank_c2(C[c1]) + rows with d before the block
template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_lf2_finish(nm_view ix, nm_lf_entry *__restrict__ lf2, uint64_t n_blocks, uint64_t n_chunks, const uint64_t *__restrict__ starts) {
    const uint64_t t = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (t >= n_chunks * 16) return;
    const uint64_t chunk = t >> 4;
    const uint32_t d = (uint32_t)(t & 15), c1 = d & 3u, c2 = d >> 2;
    const uint64_t b1 = (chunk + 1) * NM_LF2_CHUNK < n_blocks ? (chunk + 1) * NM_LF2_CHUNK : n_blocks;
    uint64_t run = starts[t] + nm_lf<BIG>(ix, c2, BIG ? ix.superC[c1] : ix.C[c1]);
    for (uint64_t b = chunk * NM_LF2_CHUNK; b < b1; b++) {
        const uint64_t c = lf2[b * 16 + d].base;
        lf2[b * 16 + d].base = run;
        run += c;
    }
}

// ---- repeat dictionary (nm_core.h): built at open, level by level from the seed table: a node = an L-mer that occurs at
// least twice; its children with at least two occurrences are the nodes of level L + 1; the nodes of level x are hashed
// into buckets of 8.
struct nm_dict_node { uint32_t klo, khi; uint64_t entry; };

// seed != nullptr: the nodes are the slots of the level-L seed table (first_slot + lane), else in[0 .. n_in)
template <bool BIG>
__global__ __launch_bounds__(NM_BLOCK) void k_dict_expand(nm_view ix, const uint64_t *__restrict__ seed, uint64_t first_slot, const nm_dict_node *__restrict__ in,
                                                          uint64_t n_in, uint32_t L, nm_dict_node *__restrict__ out, unsigned long long *__restrict__ counter,
                                                          uint64_t cap) {
    const uint64_t i = first_slot + blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    uint32_t clo[4], chi[4], n = 0;
    uint64_t cent[4];
    if (i < n_in) {
        uint32_t klo, khi;
        uint64_t entry;
        if (seed) { klo = (uint32_t)(i & ((1ULL << L) - 1ULL)); khi = (uint32_t)(i >> L); entry = seed[i]; }
        else { klo = in[i].klo; khi = in[i].khi; entry = in[i].entry; }
        if ((entry >> NM_SEED_LO_BITS) >= 2) n = nm_dict_children<BIG>(ix, klo, khi, entry, L, clo, chi, cent);
    }
    // one atomic per wave: exclusive prefix sum of n over the lanes
    uint32_t incl = n;
    const uint32_t lane = threadIdx.x & 63;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, NM_WAVE); if ((int)lane >= off) incl += v; }
    const uint32_t total = __shfl(incl, 63, NM_WAVE);
    unsigned long long base = 0;
    if (lane == 63 && total) base = atomicAdd(counter, (unsigned long long)total);
    base = __shfl(base, 63, NM_WAVE);
    const uint64_t at = base + incl - n;
    for (uint32_t c = 0; c < n; c++)
        if (at + c < cap) { nm_dict_node nd; nd.klo = clo[c]; nd.khi = chi[c]; nd.entry = cent[c]; out[at + c] = nd; }
}

__global__ __launch_bounds__(NM_BLOCK) void k_dict_insert(const nm_dict_node *__restrict__ nodes, uint64_t n, uint64_t *__restrict__ table, uint32_t bits,
                                                          unsigned int *__restrict__ fail) {
    const uint64_t i = blockIdx.x * (uint64_t)NM_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = (uint64_t)nodes[i].klo | ((uint64_t)nodes[i].khi << 32);
    const uint64_t mask = (1ULL << bits) - 1ULL;
    uint64_t b = nm_dict_bucket(key, bits);
    for (uint32_t probe = 0; probe < NM_DICT_MAX_PROBES; probe++, b = (b + 1) & mask)
        for (uint32_t j = 0; j < NM_DICT_SLOTS; j++) {
            unsigned long long *slot = (unsigned long long *)&table[(b * NM_DICT_SLOTS + j) * 2];
            if (atomicCAS(slot, (unsigned long long)NM_DICT_EMPTY, (unsigned long long)key) == (unsigned long long)NM_DICT_EMPTY) {
                table[(b * NM_DICT_SLOTS + j) * 2 + 1] = nodes[i].entry;
                return;
            }
        }
    atomicOr(fail, 1u);
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

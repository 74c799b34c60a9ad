// nm_launch.hip.h -- which kernels a segment runs, in which order, on which stream (part of nm_engine.hip).
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
    // the repeat dictionary answers for strings of dict_len bases, the second quad table for windows of quad2_m + 4: where the
    // table's window is at least as long (a 100 Mbp genome: 18 against 17) the dictionary has nothing to add to the second
    // chance (measured 239 against 220 G positions/s with it).  Elsewhere the open positions ask the second table first and
    // the dictionary takes what that leaves (k_sites phases 3, 3D): 3 Gbp, 20:200 179 -> 192 G; 24:150 295 -> 309 G
    // (`profiles/round3/ab_dictionary_order.json`)
    if (view.dict && view.quad2 && kmin >= view.quad2_m + NM_QUAD_EXT && view.dict_len <= view.quad2_m + NM_QUAD_EXT) view.dict = nullptr;
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
    // (the probes are launched whatever the handle has met so far: their blocks read a flag and return when k_sites left nothing
    //  open, 4 us of a launch; holding them back until the "open positions met" latch is visible cost the FIRST pass over a
    //  tandem-rich genome -- its launches are queued before the first of them has run -- 300 ms per 100 M positions of walks
    //  that the probes settle in 3: measured and reverted, round 4)
    if (beside) HIP_TRY(hipStreamWaitEvent(st, ix->cur->ev_join, 0));
    else if (ix->repeat_probes && (rc = nm_launch_probes<BIG>(ix, view, n, kmax, st, &probe, need)) != NM_OK) return rc;
    // what is still open: the sweep (neighbouring positions share their walks, nm_core.h) or, for A/B, one walk per position
    // (default: once the handle has met open positions -- k_resolve's latch; NM_OPT_SWEEP = 2: from the first launch on)
    const bool sweep = ix->sweep == 2 || (ix->sweep == 1 && ix->h_repeats_seen && ((volatile uint32_t *)ix->h_repeats_seen)[1] != 0);
    uint32_t *open_list = nullptr;
    if (sweep) {
        if ((rc = nm_grow(ix->cur->open_list, (NM_SWEEP_CLASSES * n_need + 1) * sizeof(uint32_t))) != NM_OK) return rc;
        open_list = (uint32_t *)ix->cur->open_list.p;
        if ((rc = nm_grow(ix->cur->need2, (n_need + 1) * sizeof(uint64_t))) != NM_OK) return rc;
    }
    const uint64_t sweep_blocks = (n_need + NM_SWEEP_BLOCK - 1) / NM_SWEEP_BLOCK;
    const dim3 wgrid((unsigned)(sweep_blocks < NM_SWEEP_MAX_BLOCKS ? sweep_blocks : NM_SWEEP_MAX_BLOCKS)), wblock(NM_SWEEP_BLOCK);
    const dim3 rgrid((unsigned)((n_need + NM_RES_WORDS - 1) / NM_RES_WORDS)), rblock(NM_RES_BLOCK);
#define NM_RES_TAIL seq_len, d_list, n_list, (const uint64_t *)hash_part, (uint32_t)sgrid.x
#define NM_LAUNCH_RES(STATS_, LIST_) do { \
        if (sweep) { nm_timed sweep_alone(ix, st, 5); \
                     hipLaunchKernelGGL((k_sweep<BIG, STATS_, LIST_>), wgrid, wblock, 0, st, view, enc, n, kmin, kmax, d_out, elem_bytes, d_status, (const uint64_t *)need, \
                                        n_need, (const uint32_t *)open_list, probe, work, NM_RES_TAIL); } \
        hipLaunchKernelGGL((k_resolve<BIG, STATS_, LIST_>), rgrid, rblock, 0, st, view, enc, n, kmin, kmax, d_out, elem_bytes, d_status, (const uint64_t *)need, n_need, \
                           probe, (const unsigned long long *)work, NM_RES_TAIL, sweep ? 1 : 0, ix->d_repeats_seen, ix->d_seen_latch, (const uint64_t *)ix->cur->need2.p); } while (0)
    {
        nm_timed timed(ix, st, 4);
        if (sweep) hipLaunchKernelGGL(k_open_words, dim3((unsigned)((n_need + NM_BLOCK * NM_OPEN_PER_LANE - 1) / (NM_BLOCK * NM_OPEN_PER_LANE))), dim3(NM_BLOCK), 0, st,
                                      (const uint64_t *)need, n_need, probe, open_list, work, (uint64_t *)ix->cur->need2.p);
        if (d_list) { if (ix->count_steps) NM_LAUNCH_RES(true, true); else NM_LAUNCH_RES(false, true); }
        else        { if (ix->count_steps) NM_LAUNCH_RES(true, false); else NM_LAUNCH_RES(false, false); }
    }
#undef NM_LAUNCH_RES
#undef NM_RES_TAIL
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

template <bool BIG, bool RC>
static void launch_fixed_k(nm_index *ix, const nm_view &view, uint64_t seq_len, uint64_t first, uint64_t num_kmers, const uint32_t *d_ks, uint32_t nk,
                           void *d_out, int elem_bytes, uint64_t *d_status, hipStream_t st) {
    const dim3 grid(nm_grid(num_kmers - first)), block(NM_BLOCK);      // positions [first, num_kmers)
    const nm_enc_word *enc = (const nm_enc_word *)ix->cur->enc.p;
    nm_timed timed(ix, st);
    if (ix->count_steps) hipLaunchKernelGGL((k_fixed_k<BIG, RC, true>), grid, block, 0, st, view, enc, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status);
    else                 hipLaunchKernelGGL((k_fixed_k<BIG, RC, false>), grid, block, 0, st, view, enc, seq_len, first, num_kmers, d_ks, nk, d_out, elem_bytes, d_status);
}

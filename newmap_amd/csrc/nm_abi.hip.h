// nm_abi.hip.h -- the C-ABI entry points of include/newmap_amd.h that search, count and guard (part of nm_engine.hip).
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
    const double t_call = nm_verbose() ? nm_now() : 0.0;
    if ((rc = nm_lane_for(ix, st)) != NM_OK) return rc;
    const double t_lane = nm_verbose() ? nm_now() : 0.0;
    struct Slow {                                                  // NEWMAP_AMD_VERBOSE: calls that held the host for long (first use of a lane)
        double t0, t1; uint64_t n;
        ~Slow() { if (t0 > 0 && nm_now() - t0 > 2e-3) fprintf(stderr, "[segment] host side of a launch of %llu positions: %.1f ms (lane %.1f ms)\n", (unsigned long long)n, (nm_now() - t0) * 1e3, (t1 - t0) * 1e3); }
    } slow{t_call, t_lane, num_kmers};
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

extern "C" uint64_t nm_dev_free_bytes(int device) {                       // 0 on error
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(device) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return 0;
    return (uint64_t)free_b;
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

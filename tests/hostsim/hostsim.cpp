// tests/hostsim/hostsim.cpp -- TEST-ONLY host simulation of the device functions in
// newmap_amd/csrc/nm_core.h.  It lets the CPU test-suite (no GPU in the build container) run the
// exact per-position logic the HIP kernels run, on an index file written by the product's host
// builder, and compare it with the oracle.  It is compiled by tests/hostsim/__init__.py into
// tests/hostsim/_build/, is not part of the package `newmap_amd`, is never loaded by it, and is not
// a fallback: the product has no CPU search path (nm_index_open refuses device < 0).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define NM_HD static inline
#include "../../newmap_amd/csrc/nm_core.h"

struct hs_index {
    nm_file_header h;
    std::vector<nm_rank_block> rank;
    std::vector<nm_strand_block> strand;
    std::vector<uint64_t> sep, seed, superC, superC2;
    std::vector<nm_rank2_block> rank2;
    std::vector<nm_lf_entry> lfb;
    std::vector<uint64_t> quad;
    nm_view v;
    bool big;
};

// host mirror of the device construction of the two-step rank blocks (nm_engine.hip: k_rank2_*)
static void hs_build_rank2(hs_index *ix) {
    const nm_view &v = ix->v;
    const uint64_t n = v.n, nblk = n / 64 + 1;
    ix->rank2.assign(nblk, nm_rank2_block{});
    std::vector<uint64_t> run(20, 0);                       // running absolute counts: 16 pairs + 4 singles
    std::vector<std::vector<uint64_t>> at_super;            // absolute counts at superblock starts
    const uint64_t per_super = 1ULL << (NM_SUPER_SHIFT - 6);
    std::vector<uint64_t> sup(20, 0);
    for (uint64_t b = 0; b < nblk; b++) {
        if (b % per_super == 0) { sup = run; at_super.push_back(run); }
        nm_rank2_block &r = ix->rank2[b];
        for (int t = 0; t < 16; t++) r.cnt2[t] = (uint32_t)(run[t] - sup[t]);
        for (int c = 0; c < 4; c++) r.cnt1[c] = (uint32_t)(run[16 + c] - sup[16 + c]);
        for (uint64_t i = b * 64; i < b * 64 + 64 && i < n; i++) {
            uint32_t c1 = 0, c2 = 0;
            const bool v1 = nm_bwt_code(v, i, c1);
            bool v2 = false;
            if (v1) {
                const uint64_t j = ix->big ? nm_lf<true>(v, c1, i) : nm_lf<false>(v, c1, i);
                v2 = nm_bwt_code(v, j, c2);
            }
            const uint64_t bit = 1ULL << (i & 63);
            if (v1) { r.valid1 |= bit; if (c1 & 1) r.c1lo |= bit; if (c1 & 2) r.c1hi |= bit; run[16 + c1]++; }
            if (v2) { r.valid2 |= bit; if (c2 & 1) r.c2lo |= bit; if (c2 & 2) r.c2hi |= bit; run[c1 * 4 + c2]++; }
        }
    }
    ix->superC2.assign(at_super.size() * 16, 0);
    for (size_t sb = 0; sb < at_super.size(); sb++)
        for (uint32_t x = 0; x < 4; x++)
            for (uint32_t y = 0; y < 4; y++) {
                // first row of the suffixes starting "y x": LF_y of the first row starting with x
                const uint64_t base = ix->big ? nm_lf<true>(v, y, v.C[x]) : nm_lf<false>(v, y, v.C[x]);
                ix->superC2[sb * 16 + x * 4 + y] = base + at_super[sb][x * 4 + y];
            }
    ix->v.rank2 = ix->rank2.data();
    ix->v.superC2 = ix->superC2.data();
}

static void hs_encode(const uint8_t *seq, uint64_t seq_len, std::vector<nm_enc_word> &enc) {
    const uint64_t n_words = seq_len / 64 + 3;
    enc.assign(n_words, nm_enc_word{0, 0, 0, 0});
    for (uint64_t pos = 0; pos < n_words * 64; pos++) {
        uint32_t code = pos < seq_len ? nm_base_code(seq[pos]) : 4;
        nm_enc_word &w = enc[pos >> 6];
        const uint64_t bit = 1ULL << (pos & 63);
        if (code > 3) w.amb |= bit;
        else { if (code & 1) w.lo |= bit; if (code & 2) w.hi |= bit; }
    }
}

extern "C" {

hs_index *hs_open(const char *path, int seed_len_override, int force_big) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return nullptr;
    hs_index *ix = new hs_index();
    bool ok = fread(&ix->h, sizeof ix->h, 1, fp) == 1 && memcmp(ix->h.magic, NM_MAGIC, 8) == 0;
    if (ok) {
        ix->rank.resize(ix->h.n_rank_blocks);
        ix->strand.resize(ix->h.n_strand_blocks);
        ix->sep.resize(ix->h.n_sep);
        ok = fread(ix->rank.data(), sizeof(nm_rank_block), ix->rank.size(), fp) == ix->rank.size() &&
             fread(ix->strand.data(), sizeof(nm_strand_block), ix->strand.size(), fp) == ix->strand.size() &&
             fread(ix->sep.data(), 8, ix->sep.size(), fp) == ix->sep.size();
    }
    fclose(fp);
    if (!ok) { delete ix; return nullptr; }
    const nm_file_header &h = ix->h;
    uint64_t C[4];
    C[0] = h.n_sep;
    for (int c = 1; c < 4; c++) C[c] = C[c - 1] + h.base_count[c - 1];
    ix->superC.resize(h.n_super * 4);
    for (uint64_t j = 0; j < h.n_super; j++)
        for (int c = 0; c < 4; c++) ix->superC[j * 4 + c] = C[c] + h.super_cnt[j][c];
    ix->big = force_big || h.n_super > 1;
    nm_view &v = ix->v;
    v.rank = ix->rank.data(); v.strand = ix->strand.data(); v.sep = ix->sep.data();
    v.seed = nullptr; v.superC = ix->superC.data(); v.n = h.n; v.n_sep = h.n_sep;
    for (int c = 0; c < 4; c++) v.C[c] = C[c];
    v.seed_len = 0; v.n_super = (uint32_t)h.n_super; v.seed_policy = 0; v.pair_m = 0; v.pair = nullptr; v.rank2 = nullptr; v.superC2 = nullptr; v.lfb = nullptr; v.quad = nullptr; v.quad_m = 0;
    uint32_t s = seed_len_override < 0 ? h.seed_len : (uint32_t)seed_len_override;
    if (s > 12) s = 12;                     // keep the simulated table small
    if (s && h.n >= 2) {
        ix->seed.resize(1ULL << (2 * s));
        for (uint64_t slot = 0; slot < ix->seed.size(); slot++)
            ix->seed[slot] = ix->big ? nm_seed_entry<true>(v, slot, s) : nm_seed_entry<false>(v, slot, s);
        v.seed = ix->seed.data();
        v.seed_len = s;
    }
    return ix;
}

void hs_close(hs_index *ix) { delete ix; }
// LF blocks (one 16-byte entry per block and base), host mirror of k_lf_blocks
void hs_enable_lfb(hs_index *ix, int on) {
    if (on && ix->lfb.empty()) {
        const uint64_t nb = ix->v.n / 64 + 1;
        ix->lfb.resize(nb * 4);
        ix->v.lfb = nullptr;
        for (uint64_t b = 0; b < nb; b++) {
            if (ix->big) nm_lf_entries_of_block<true>(ix->v, b, &ix->lfb[b * 4]);
            else nm_lf_entries_of_block<false>(ix->v, b, &ix->lfb[b * 4]);
        }
    }
    ix->v.lfb = on ? ix->lfb.data() : nullptr;
}
// LF blocks against the packed rank blocks at every row
uint64_t hs_check_lfb(hs_index *ix) {
    nm_view packed = ix->v;
    packed.lfb = nullptr;
    uint64_t bad = 0;
    for (uint64_t i = 0; i <= ix->v.n; i++)
        for (uint32_t c = 0; c < 4; c++) {
            const uint64_t a = ix->big ? nm_lf<true>(ix->v, c, i) : nm_lf<false>(ix->v, c, i);
            const uint64_t b = ix->big ? nm_lf<true>(packed, c, i) : nm_lf<false>(packed, c, i);
            if (a != b) bad++;
        }
    return bad;
}
void hs_enable_rank2(hs_index *ix, int on) {
    if (on && ix->rank2.empty()) hs_build_rank2(ix);
    ix->v.rank2 = on ? ix->rank2.data() : nullptr;
    ix->v.superC2 = on ? ix->superC2.data() : nullptr;
}
// two-step blocks against the one-step structure: LF_x and LF_y(LF_x) at every row and base pair
uint64_t hs_check_rank2(hs_index *ix, uint64_t stride) {
    const nm_view &v = ix->v;
    uint64_t bad = 0;
    for (uint64_t i = 0; i <= v.n; i += stride)
        for (uint32_t x = 0; x < 4; x++) {
            const uint64_t want1 = ix->big ? nm_lf<true>(v, x, i) : nm_lf<false>(v, x, i);
            if (nm_lf1_r2(v, x, i) != want1) bad++;
            for (uint32_t y = 0; y < 4; y++) {
                uint64_t o1, o2;
                nm_lf12(v, x, y, i, o1, o2);
                const uint64_t want2 = ix->big ? nm_lf<true>(v, y, want1) : nm_lf<false>(v, y, want1);
                if (o1 != want1 || o2 != want2) bad++;
            }
        }
    return bad;
}

// pair-table entries must equal the plain seed entries of the (m+1)-mers they stand for
uint64_t hs_check_pair(hs_index *ix, uint32_t m) {
    const uint32_t s = m + 1;
    const uint64_t mask = (1ULL << m) - 1;
    uint64_t bad = 0;
    for (uint64_t slot = 0; slot < (1ULL << (2 * m)); slot++) {
        const uint64_t ylo = slot & mask, yhi = slot >> m;
        for (uint32_t e = 0; e < 8; e++) {
            uint64_t sl;
            if (e < 4) sl = (uint64_t)(e & 1) | (ylo << 1) | ((uint64_t)(e >> 1) << s) | (yhi << (s + 1));
            else { const uint32_t b = e - 4; sl = ylo | ((uint64_t)(b & 1) << m) | (yhi << s) | ((uint64_t)(b >> 1) << (s + m)); }
            if (sl != nm_pair_seed_slot(slot, m, e)) bad++;
            const uint64_t want = ix->big ? nm_seed_entry<true>(ix->v, sl, s) : nm_seed_entry<false>(ix->v, sl, s);
            const uint64_t got = ix->big ? nm_pair_entry<true>(ix->v, slot, m, e) : nm_pair_entry<false>(ix->v, slot, m, e);
            // an empty interval may sit anywhere: compare sizes, and starts only when non-empty
            const uint64_t wc = want >> NM_SEED_LO_BITS, gc = got >> NM_SEED_LO_BITS;
            if (wc != gc || (wc && want != got)) bad++;
        }
    }
    return bad;
}
// quad table (nm_core.h: nm_quad_build_one, the body of k_quad_build) from the simulated seed table; then
// every bit a group of four positions can read is compared with a direct count of its (m+3)-mer.
// Returns the number of wrong bits (+ 2^32 per 16-bit piece no lane wrote).
uint64_t hs_check_quad(hs_index *ix) {
    const uint32_t m = ix->v.seed_len, w = m + 3;
    if (!m || m > 8) return ~0ULL;
    const uint64_t cores = 1ULL << (2 * m);
    ix->quad.assign(cores * 4, 0x5A5A5A5A5A5A5A5AULL);
    std::vector<uint64_t> ref(cores * 4, 0x5A5A5A5A5A5A5A5AULL), other(cores * 4, 0xA5A5A5A5A5A5A5A5ULL);
    for (uint64_t Z = 0; Z < cores; Z++) {
        if (ix->big) { nm_quad_build_one<true>(ix->v, Z, m, ix->quad.data()); nm_quad_build_one<true>(ix->v, Z, m, other.data()); }
        else { nm_quad_build_one<false>(ix->v, Z, m, ix->quad.data()); nm_quad_build_one<false>(ix->v, Z, m, other.data()); }
    }
    uint64_t bad = 0;
    for (uint64_t i = 0; i < cores * 4; i++)                // a piece nobody wrote keeps its (different) fill pattern
        for (int h = 0; h < 4; h++)
            if (((ix->quad[i] >> (16 * h)) & 0xFFFF) != ((other[i] >> (16 * h)) & 0xFFFF)) bad += 1ULL << 32;
    ix->v.quad = ix->quad.data();
    ix->v.quad_m = m;
    // windows of m + 6 bases: L (3) . core (m) . R (3); position i of the group reads word i
    const uint64_t n_win = 1ULL << (2 * (m + 6));
    for (uint64_t x = 0; x < n_win; x++) {
        nm_window win{0, 0, 0};
        for (uint32_t j = 0; j < m + 6; j++) {
            const uint32_t c = (uint32_t)(x >> (2 * j)) & 3u;
            win.lo |= (uint64_t)(c & 1u) << j;
            win.hi |= (uint64_t)(c >> 1) << j;
        }
        const uint64_t slot = nm_quad_slot(win, m);
        const uint32_t got = nm_quad_bits(win, m, ix->quad.data() + slot * 4);
        for (uint32_t i = 0; i < 4; i++) {
            nm_window wi{win.lo >> i, win.hi >> i, 0};
            const uint64_t e = ix->big ? nm_seed_entry<true>(ix->v, nm_seed_slot(wi, w), w) : nm_seed_entry<false>(ix->v, nm_seed_slot(wi, w), w);
            const uint32_t want = (e >> NM_SEED_LO_BITS) == 1 ? 1u : 0u;
            if (((got >> i) & 1u) != want) bad++;
        }
    }
    return bad;
}
// level-wise seed construction must reproduce the entry-by-entry one
uint64_t hs_check_levels(hs_index *ix, uint32_t s) {
    uint64_t bad = 0;
    std::vector<uint64_t> parent(1ULL << (2 * (s - 1)));
    for (uint64_t slot = 0; slot < parent.size(); slot++)
        parent[slot] = ix->big ? nm_seed_entry<true>(ix->v, slot, s - 1) : nm_seed_entry<false>(ix->v, slot, s - 1);
    for (uint64_t slot = 0; slot < (1ULL << (2 * s)); slot++) {
        const uint64_t p = parent[nm_seed_parent_slot(slot, s)];
        const uint64_t got = ix->big ? nm_seed_entry_from_parent<true>(ix->v, p, slot, s) : nm_seed_entry_from_parent<false>(ix->v, p, slot, s);
        const uint64_t want = ix->big ? nm_seed_entry<true>(ix->v, slot, s) : nm_seed_entry<false>(ix->v, slot, s);
        if (got != want) bad++;
    }
    return bad;
}

uint64_t hs_info(hs_index *ix, int what) {
    switch (what) { case 0: return ix->h.n; case 1: return ix->h.n_fwd; case 2: return ix->h.n_sep;
                    case 3: return ix->h.n_records; case 4: return ix->h.raw_bases; case 5: return ix->v.seed_len;
                    default: return 0; }
}

// returns 0 ok, 8 k-mer not found (like NM_E_KMER_NOT_FOUND)
int hs_min_unique(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmin,
                  uint32_t kmax, int use_rc, int elem_bytes, void *out, uint64_t *status) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    for (int i = 0; i < 8; i++) status[i] = 0;
    status[2] = ~0ULL;
    for (uint64_t p = 0; p < num_kmers; p++) {
        bool amb0 = false, err = false;
        nm_tally t = {0, 0, 0, 0};
        uint32_t r;
        if (ix->big) r = use_rc ? nm_min_unique_one<true, true>(ix->v, enc.data(), p, kmin, kmax, amb0, err, t)
                                : nm_min_unique_one<true, false>(ix->v, enc.data(), p, kmin, kmax, amb0, err, t);
        else         r = use_rc ? nm_min_unique_one<false, true>(ix->v, enc.data(), p, kmin, kmax, amb0, err, t)
                                : nm_min_unique_one<false, false>(ix->v, enc.data(), p, kmin, kmax, amb0, err, t);
        if (elem_bytes == 1) ((uint8_t *)out)[p] = (uint8_t)r;
        else if (elem_bytes == 2) ((uint16_t *)out)[p] = (uint16_t)r;
        else ((uint32_t *)out)[p] = r;
        status[0] += amb0;
        if (err) { status[1] = 1; if (p < status[2]) status[2] = p; }
        status[3] += t.steps; status[4] += t.blocks; status[5] += t.seeds; status[6] += t.strands;
        status[7] += !amb0;
    }
    return status[1] ? 8 : 0;
}

// nm_base_codes4 (the encode kernel's word-wide classification) against nm_base_code on every byte value in every
// byte position with pseudo-random neighbours; returns the number of disagreements
uint64_t hs_check_codes4(uint64_t rounds) {
    uint64_t bad = 0, state = 0x9E3779B97F4A7C15ULL;
    for (uint64_t r = 0; r < rounds; r++)
        for (uint32_t pos = 0; pos < 4; pos++)
            for (uint32_t v = 0; v < 256; v++) {
                state = state * 6364136223846793005ULL + 1442695040888963407ULL;
                uint32_t x = (uint32_t)(state >> 32);
                x = (x & ~(0xFFu << (8 * pos))) | (v << (8 * pos));
                uint32_t lo, hi, amb;
                nm_base_codes4(x, lo, hi, amb);
                for (uint32_t i = 0; i < 4; i++) {
                    const uint32_t c = nm_base_code((x >> (8 * i)) & 0xFFu);
                    const uint32_t wl = c < 4 ? (c & 1u) : 0u, wh = c < 4 ? (c >> 1) : 0u, wa = c > 3;
                    if (((lo >> i) & 1u) != wl || ((hi >> i) & 1u) != wh || ((amb >> i) & 1u) != wa) bad++;
                }
            }
    return bad;
}

// the repeat probes of one segment (k_repeat_probe): the probe word of every stride, and what the consumers
// make of them: decided[p] = the element nm_probe_kstar / nm_probe_element give position p, 0xFFFFFFFF where the
// probes leave it open.  Returns the LF steps spent.
uint64_t hs_repeat_probes(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmin,
                          uint32_t kmax, uint32_t stride, uint32_t coarse_stride, uint32_t *words, uint32_t *decided) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    uint64_t steps = 0;
    const uint64_t n_probes = (num_kmers + stride - 1) / stride;
    // coarse probes first (k_repeat_probe_coarse; coarse_stride 0 = none): strides they settle completely get their
    // word without a walk (k_repeat_probe)
    std::vector<uint32_t> coarse;
    if (coarse_stride) {
        coarse.resize((num_kmers + coarse_stride - 1) / coarse_stride);
        for (uint64_t c = 0; c < coarse.size(); c++) {
            nm_tally t = {0, 0, 0, 0};
            uint32_t settled, exact;
            if (ix->big) nm_repeat_probe_ex<true>(ix->v, enc.data(), c * coarse_stride, kmax, coarse_stride, t, settled, exact);
            else nm_repeat_probe_ex<false>(ix->v, enc.data(), c * coarse_stride, kmax, coarse_stride, t, settled, exact);
            coarse[c] = settled;
            steps += t.steps;
        }
    }
    for (uint64_t j = 0; j < n_probes; j++) {
        nm_tally t = {0, 0, 0, 0};
        const uint64_t P = j * stride;
        if (coarse_stride && nm_coarse_covers(coarse[P / coarse_stride], (uint32_t)(P % coarse_stride), stride)) { words[j] = stride; continue; }
        words[j] = ix->big ? nm_repeat_probe<true>(ix->v, enc.data(), P, kmax, stride, t)
                           : nm_repeat_probe<false>(ix->v, enc.data(), P, kmax, stride, t);
        steps += t.steps;
    }
    words[n_probes] = 0;
    for (uint64_t p = 0; p < num_kmers; p++) {
        const uint32_t ks = nm_probe_kstar(words[p / stride], words[p / stride + 1], (uint32_t)(p % stride), stride, kmax);
        if (ks == NM_PROBE_OPEN) { decided[p] = 0xFFFFFFFFu; continue; }
        nm_window w = nm_load_window(enc.data(), p);
        uint32_t kbase = 0;
        decided[p] = nm_probe_element(ks, kmin, kmax, ks < kmin && nm_all_valid(enc.data(), p, w, kbase, 0, kmin));
    }
    return steps;
}

int hs_fixed_k(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks,
               uint32_t nk, int use_rc, int elem_bytes, void *out, uint64_t *status) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    for (int i = 0; i < 8; i++) status[i] = 0;
    status[2] = ~0ULL;
    for (uint64_t p = 0; p < num_kmers; p++) {
        bool amb0 = false, err = false;
        nm_tally t = {0, 0, 0, 0};
        uint32_t r;
        if (ix->big) r = use_rc ? nm_fixed_k_one<true, true>(ix->v, enc.data(), p, seq_len, ks, nk, amb0, err, t)
                                : nm_fixed_k_one<true, false>(ix->v, enc.data(), p, seq_len, ks, nk, amb0, err, t);
        else         r = use_rc ? nm_fixed_k_one<false, true>(ix->v, enc.data(), p, seq_len, ks, nk, amb0, err, t)
                                : nm_fixed_k_one<false, false>(ix->v, enc.data(), p, seq_len, ks, nk, amb0, err, t);
        if (elem_bytes == 1) ((uint8_t *)out)[p] = (uint8_t)r;
        else if (elem_bytes == 2) ((uint16_t *)out)[p] = (uint16_t)r;
        else ((uint32_t *)out)[p] = r;
        status[0] += amb0;
        if (err) { status[1] = 1; if (p < status[2]) status[2] = p; }
    }
    return status[1] ? 8 : 0;
}

void hs_count(hs_index *ix, const uint8_t *seq, const uint64_t *starts, const uint64_t *lens, uint64_t n, uint32_t *out) {
    for (uint64_t q = 0; q < n; q++) {
        nm_tally t = {0, 0, 0, 0};
        out[q] = ix->big ? nm_count_fwd_one<true>(ix->v, seq + starts[q], lens[q], t)
                         : nm_count_fwd_one<false>(ix->v, seq + starts[q], lens[q], t);
    }
}

// several sequences in lock-step x several indexes; range mode when nk == 0
int hs_multi(hs_index **ixs, uint32_t n_idx, const uint8_t **seqs, uint32_t n_seq, uint64_t seq_len,
             uint64_t num_kmers, uint32_t kmin, uint32_t kmax, const uint32_t *ks, uint32_t nk, int use_rc,
             int elem_bytes, void *out, uint64_t *status) {
    nm_multi_args a;
    std::vector<std::vector<nm_enc_word>> enc(n_seq);
    a.n_idx = n_idx; a.n_seq = n_seq;
    for (uint32_t f = 0; f < n_idx; f++) { a.view[f] = ixs[f]->v; a.view[f].seed = nullptr; a.view[f].seed_len = 0; }
    for (uint32_t i = 0; i < n_seq; i++) { hs_encode(seqs[i], seq_len, enc[i]); a.enc[i] = enc[i].data(); }
    for (int i = 0; i < 8; i++) status[i] = 0;
    status[2] = ~0ULL;
    for (uint64_t p = 0; p < num_kmers; p++) {
        bool amb0 = false, err = false;
        uint32_t r;
        if (nk == 0) r = use_rc ? nm_min_unique_multi_one<true>(a, p, kmin, kmax, amb0, err)
                                : nm_min_unique_multi_one<false>(a, p, kmin, kmax, amb0, err);
        else         r = use_rc ? nm_fixed_k_multi_one<true>(a, p, seq_len, ks, nk, amb0, err)
                                : nm_fixed_k_multi_one<false>(a, p, seq_len, ks, nk, amb0, err);
        if (elem_bytes == 1) ((uint8_t *)out)[p] = (uint8_t)r;
        else if (elem_bytes == 2) ((uint16_t *)out)[p] = (uint16_t)r;
        else ((uint32_t *)out)[p] = r;
        status[0] += amb0;
        if (err) { status[1] = 1; if (p < status[2]) status[2] = p; }
    }
    return status[1] ? 8 : 0;
}

void hs_upper(const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmax, uint32_t *out) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    for (uint64_t p = 0; p < num_kmers; p++) out[p] = nm_upper_one(enc.data(), p, kmax);
}

}  // extern "C"

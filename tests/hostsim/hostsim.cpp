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
#include "../../newmap_amd/csrc/nm_fasta_scan.hpp"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct hs_index {
    nm_file_header h;
    std::vector<nm_rank_block> rank;
    std::vector<nm_strand_block> strand;
    std::vector<uint64_t> sep, seed, superC;
    std::vector<nm_lf_entry> lfb, lf2;
    std::vector<uint64_t> quad, quad2, dict;
    std::vector<uint8_t> lcp;
    nm_view v;
    bool big;
};

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
    if (ok && ix->h.off_lcp) {                                // LCP bytes (optional section)
        ix->lcp.resize(ix->h.n + 1 + 64);                     // (+ slack: the sweep reads 32 bytes at a time)
        ok = fseeko(fp, (off_t)ix->h.off_lcp, SEEK_SET) == 0 && fread(ix->lcp.data(), 1, ix->h.n + 1, fp) == ix->h.n + 1;
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
    v.seed_len = 0; v.n_super = (uint32_t)h.n_super; v.seed_policy = 0; v.lfb = nullptr; v.lf2 = nullptr; v.dict = nullptr; v.dict_len = v.dict_bits = 0; v.lcp = nullptr; v.quad = nullptr; v.quad_m = 0; v.quad2 = nullptr; v.quad2_m = 0;
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
// two-base LF blocks: host mirror of k_lf2_bits / k_lf2_chunk_sums / k_lf2_finish (nm_tables.hip.h: nm_build_lf2)
void hs_enable_lf2(hs_index *ix, int on) {
    if (on && ix->lf2.empty()) {
        const nm_view v = ix->v;
        const uint64_t nb = v.n / 64 + 1;
        ix->lf2.assign(nb * 16, nm_lf_entry{0, 0});
        for (uint64_t b = 0; b < nb; b++)
            for (uint32_t lane = 0; lane < 64; lane++) {
                const uint32_t d = ix->big ? nm_bwt2_code<true>(v, b * 64 + lane) : nm_bwt2_code<false>(v, b * 64 + lane);
                if (d < 16) { ix->lf2[b * 16 + d].bits |= 1ULL << lane; ix->lf2[b * 16 + d].base++; }
            }
        for (uint32_t d = 0; d < 16; d++) {
            const uint32_t c1 = d & 3u, c2 = d >> 2;
            uint64_t run = ix->big ? nm_lf<true>(v, c2, v.superC[c1]) : nm_lf<false>(v, c2, v.C[c1]);
            for (uint64_t b = 0; b < nb; b++) { const uint64_t c = ix->lf2[b * 16 + d].base; ix->lf2[b * 16 + d].base = run; run += c; }
        }
    }
    ix->v.lf2 = on ? ix->lf2.data() : nullptr;
}
// the two-base step against two single steps, at every row and dinucleotide
uint64_t hs_check_lf2(hs_index *ix) {
    uint64_t bad = 0;
    if (!ix->v.lf2) return ~0ULL;
    for (uint64_t i = 0; i <= ix->v.n; i++)
        for (uint32_t d = 0; d < 16; d++) {
            const uint32_t c1 = d & 3u, c2 = d >> 2;
            uint64_t lo = i, hi = i;
            if (ix->big) nm_lf2_interval<true>(ix->v, c1, c2, lo, hi); else nm_lf2_interval<false>(ix->v, c1, c2, lo, hi);
            const uint64_t one = ix->big ? nm_lf<true>(ix->v, c2, nm_lf<true>(ix->v, c1, i)) : nm_lf<false>(ix->v, c2, nm_lf<false>(ix->v, c1, i));
            if (lo != one || hi != one) bad++;
        }
    return bad;
}
// both-direction steps (nm_bi_extend) against two independent backward searches: strings grown base by base to the left and
// to the right, following the text `seq` around a random position (so that they occur) or at random; returns the number of
// disagreements
uint64_t hs_check_bi(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t rounds, uint64_t rng) {
    const nm_view &v = ix->v;
    uint64_t bad = 0;
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    auto search = [&](const std::vector<uint32_t> &x, uint64_t &lo, uint64_t &hi) {      // rows of x by a plain backward search
        lo = 0; hi = v.n;
        for (size_t i = x.size(); i-- > 0 && lo < hi;) {
            if (ix->big) nm_lf_interval<true>(v, x[i], lo, hi); else nm_lf_interval<false>(v, x[i], lo, hi);
        }
        if (hi < lo) hi = lo;
    };
    for (uint64_t r = 0; r < rounds; r++) {
        std::vector<uint32_t> x;
        const bool follow = seq_len > 0 && (r & 3) != 0;
        uint64_t a = follow ? next() % seq_len : 0, b = a;          // the string is seq[a .. b) when it follows the text
        nm_bi iv = {0, 0, v.n};
        nm_tally t = {0, 0, 0, 0};
        for (uint32_t step = 0; step < 48 && iv.s; step++) {
            bool left = next() & 1;
            uint32_t c = (uint32_t)(next() & 3);
            if (follow) {
                if (left && a == 0) left = false;
                if (!left && b >= seq_len) left = true;
                if (left && a == 0) break;
                const uint32_t code = nm_base_code(seq[left ? a - 1 : b]);
                if (code > 3) break;
                c = code;
                if (left) a--; else b++;
            }
            if (left) {
                if (ix->big) nm_bi_extend<true>(v, iv, c, t); else nm_bi_extend<false>(v, iv, c, t);
                x.insert(x.begin(), c);
            } else {
                nm_bi y = {iv.l, iv.k, iv.s};
                if (ix->big) nm_bi_extend<true>(v, y, 3u - c, t); else nm_bi_extend<false>(v, y, 3u - c, t);
                iv.k = y.l; iv.l = y.k; iv.s = y.s;
                x.push_back(c);
            }
            std::vector<uint32_t> rc(x.size());
            for (size_t i = 0; i < x.size(); i++) rc[i] = 3u - x[x.size() - 1 - i];
            uint64_t lo, hi, rlo, rhi;
            search(x, lo, hi);
            search(rc, rlo, rhi);
            if (hi - lo != iv.s || rhi - rlo != iv.s) { bad++; break; }
            if (iv.s && (lo != iv.k || rlo != iv.l)) { bad++; break; }
            if (follow && iv.s == 0) { bad++; break; }            // a string of the text occurs
        }
    }
    return bad;
}
static int g_sweep = 0;
void hs_set_sweep(int on) { g_sweep = on; }
// the sweep reads the index file's LCP bytes (on) or walks where the end of a chain moves; returns 0 if the file has none
int hs_enable_lcp(hs_index *ix, int on) { ix->v.lcp = on && !ix->lcp.empty() ? ix->lcp.data() : nullptr; return ix->lcp.empty() ? 0 : 1; }
// lcp[row] of the index file against a direct comparison of the two suffixes' k-mers is left to the Python side: the bytes
int64_t hs_lcp_byte(hs_index *ix, uint64_t row) { return row < ix->lcp.size() && !ix->lcp.empty() ? (int64_t)ix->lcp[row] : -1; }
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
// quad table (nm_core.h: nm_quad_build_one, the body of k_quad_build) from the simulated seed table; then
// every bit a group of four positions can read is compared with a direct count of its (m+3)-mer.
// Returns the number of wrong bits (+ 2^32 per 16-bit piece no lane wrote).
uint64_t hs_check_quad(hs_index *ix) {
    const uint32_t m = ix->v.seed_len, w = m + NM_QUAD_EXT;
    if (!m || m > 8) return ~0ULL;
    const uint64_t cores = 1ULL << (2 * m);
    ix->quad.assign(cores * NM_QUAD_WORDS, 0x5A5A5A5A5A5A5A5AULL);
    std::vector<uint64_t> other(cores * NM_QUAD_WORDS, 0xA5A5A5A5A5A5A5A5ULL);
    for (uint64_t Z = 0; Z < cores; Z++) {
        if (ix->big) { nm_quad_build_one<true>(ix->v, Z, m, ix->quad.data()); nm_quad_build_one<true>(ix->v, Z, m, other.data()); }
        else { nm_quad_build_one<false>(ix->v, Z, m, ix->quad.data()); nm_quad_build_one<false>(ix->v, Z, m, other.data()); }
    }
    uint64_t bad = 0;
    for (uint64_t i = 0; i < cores * NM_QUAD_WORDS; i++)    // a 16-bit piece nobody wrote keeps its (different) fill pattern
        for (int h = 0; h < 4; h++)
            if (((ix->quad[i] >> (16 * h)) & 0xFFFF) != ((other[i] >> (16 * h)) & 0xFFFF)) bad += 1ULL << 32;
    ix->v.quad = ix->quad.data();
    ix->v.quad_m = m;
    // stretches of m + 8 bases: L (4) . core (m) . R (4); the windows 0, 1, 3, 4 bases in are read from one entry.
    // A pseudo-random sample of 4 M of them (all, if there are fewer).
    const uint32_t span = m + 8;
    const uint64_t n_all = 1ULL << (2 * span), n_win = n_all <= (1ULL << 22) ? n_all : (1ULL << 22);
    uint64_t state = 0x9E3779B97F4A7C15ULL;
    for (uint64_t k = 0; k < n_win; k++) {
        uint64_t x = k;
        if (n_win != n_all) { state = state * 6364136223846793005ULL + 1442695040888963407ULL; x = (state >> 11) & (n_all - 1); }
        nm_window win{0, 0, 0};
        for (uint32_t j = 0; j < span; j++) {
            const uint32_t c = (uint32_t)(x >> (2 * j)) & 3u;
            win.lo |= (uint64_t)(c & 1u) << j;
            win.hi |= (uint64_t)(c >> 1) << j;
        }
        const uint64_t *entry = ix->quad.data() + nm_quad_slot(win, m) * NM_QUAD_WORDS;
        uint32_t b[4];
        nm_quad_index(win, m, b);
        const uint64_t *p01 = nm_quad_pair01(entry, b), *p34 = nm_quad_pair34(entry, b);
        const uint64_t e[4] = {p01[0], p01[1], p34[0], p34[1]};
        const uint32_t got = nm_quad_bits(b, e);
        if (got & ~NM_QUAD_OFFSETS) bad++;
        for (uint32_t i = 0; i < 5; i++) {
            if (!((NM_QUAD_OFFSETS >> i) & 1u)) continue;
            nm_window wi{win.lo >> i, win.hi >> i, 0};
            const uint64_t se = ix->big ? nm_seed_entry<true>(ix->v, nm_seed_slot(wi, w), w) : nm_seed_entry<false>(ix->v, nm_seed_slot(wi, w), w);
            const uint32_t want = (se >> NM_SEED_LO_BITS) == 1 ? 1u : 0u;
            if (((got >> i) & 1u) != want) bad++;
        }
        if (nm_quad_once_first(ix->v, win, 1000) != ((got & 1u) != 0)) bad++;
    }
    return bad;
}
// a second quad table with longer cores (k_resolve's second chance), from a temporary seed table of that length
int hs_build_quad2(hs_index *ix, uint32_t m2) {
    if (m2 < 3 || m2 > 9) return -1;
    const uint64_t cores = 1ULL << (2 * m2);
    std::vector<uint64_t> seed(cores);
    nm_view v = ix->v;
    for (uint64_t slot = 0; slot < cores; slot++)
        seed[slot] = ix->big ? nm_seed_entry<true>(v, slot, m2) : nm_seed_entry<false>(v, slot, m2);
    v.seed = seed.data();
    v.seed_len = m2;
    ix->quad2.assign(cores * NM_QUAD_WORDS, 0);
    for (uint64_t Z = 0; Z < cores; Z++) {
        if (ix->big) nm_quad_build_one<true>(v, Z, m2, ix->quad2.data());
        else nm_quad_build_one<false>(v, Z, m2, ix->quad2.data());
    }
    ix->v.quad2 = ix->quad2.data();
    ix->v.quad2_m = m2;
    return 0;
}
// the repeat dictionary (nm_core.h; nm_tables.hip.h: nm_build_dict): level by level from a seed level of length s0 with the
// same node expansion (nm_dict_children), inserted serially with the kernels' bucket rule.  Returns the number of strings,
// or -1 if a bucket chain ran over.
int64_t hs_build_dict(hs_index *ix, uint32_t s0, uint32_t x) {
    if (x <= s0 || x > NM_DICT_MAX_LEN || s0 < 1 || s0 > 10) return -1;
    struct Node { uint32_t klo, khi; uint64_t entry; };
    std::vector<Node> cur, nxt;
    nm_view v = ix->v;
    for (uint64_t slot = 0; slot < (1ULL << (2 * s0)); slot++) {
        const uint64_t e = ix->big ? nm_seed_entry<true>(v, slot, s0) : nm_seed_entry<false>(v, slot, s0);
        if ((e >> NM_SEED_LO_BITS) >= 2) cur.push_back({(uint32_t)(slot & ((1ULL << s0) - 1)), (uint32_t)(slot >> s0), e});
    }
    for (uint32_t L = s0; L < x; L++) {
        nxt.clear();
        for (const Node &nd : cur) {
            uint32_t lo[4], hi[4];
            uint64_t en[4];
            const uint32_t n = ix->big ? nm_dict_children<true>(v, nd.klo, nd.khi, nd.entry, L, lo, hi, en) : nm_dict_children<false>(v, nd.klo, nd.khi, nd.entry, L, lo, hi, en);
            for (uint32_t c = 0; c < n; c++) nxt.push_back({lo[c], hi[c], en[c]});
        }
        cur.swap(nxt);
    }
    uint32_t bits = 4;
    while ((1ULL << bits) * 4 < cur.size()) bits++;
    ix->dict.assign((16ULL << bits), NM_DICT_EMPTY);
    const uint64_t mask = (1ULL << bits) - 1;
    for (const Node &nd : cur) {
        const uint64_t key = (uint64_t)nd.klo | ((uint64_t)nd.khi << 32);
        uint64_t b = nm_dict_bucket(key, bits);
        bool done = false;
        for (uint32_t probe = 0; probe < NM_DICT_MAX_PROBES && !done; probe++, b = (b + 1) & mask)
            for (uint32_t j = 0; j < NM_DICT_SLOTS && !done; j++)
                if (ix->dict[(b * NM_DICT_SLOTS + j) * 2] == NM_DICT_EMPTY) { ix->dict[(b * NM_DICT_SLOTS + j) * 2] = key; ix->dict[(b * NM_DICT_SLOTS + j) * 2 + 1] = nd.entry; done = true; }
        if (!done) return -1;
    }
    ix->v.dict = ix->dict.data();
    ix->v.dict_len = x;
    ix->v.dict_bits = bits;
    return (int64_t)cur.size();
}
// what the dictionary says about a raw string of dict_len bases: -1 not in it, else its interval size (saturating)
int64_t hs_dict_lookup(hs_index *ix, const uint8_t *kmer) {
    std::vector<nm_enc_word> enc;
    hs_encode(kmer, ix->v.dict_len, enc);
    const nm_window w = nm_load_window(enc.data(), 0);
    uint64_t e;
    if (!nm_dict_find(ix->v, nm_dict_key(w, ix->v.dict_len), e)) return -1;
    return (int64_t)(e >> NM_SEED_LO_BITS);
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
// k_segment_hash / the fingerprint stage of k_sites: the fingerprint of positions [0, end) of a segment from its encoded words
uint64_t hs_segment_hash(const uint8_t *seq, uint64_t seq_len, uint64_t end) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    std::vector<uint64_t> tab(NM_HASH_TAB_WORDS);
    nm_hash_fill_tables(tab.data());
    uint64_t h = 0;
    for (uint64_t w = 0; w * 64 < end && w < enc.size(); w++) h += nm_hash_segment_word(tab.data(), enc[w], w, end);
    return h;
}

// nm_guard_range_one / nm_guard_list_one over a segment: returns the first position for which the reference would raise, or ~0
uint64_t hs_guard(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks, uint32_t nk, int range_mode,
                  uint32_t initial_len, int use_rc) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    nm_view v = ix->v;
    uint32_t kmin = ks[0], kmax = ks[0];
    for (uint32_t i = 1; i < nk; i++) { if (ks[i] < kmin) kmin = ks[i]; if (ks[i] > kmax) kmax = ks[i]; }
    for (uint64_t p = 0; p < num_kmers; p++) {
        nm_tally t = {0, 0, 0, 0};
        bool bad;
        if (range_mode) bad = use_rc ? nm_guard_range_one<true, true>(v, enc.data(), p, kmin, kmax, initial_len, t) : nm_guard_range_one<true, false>(v, enc.data(), p, kmin, kmax, initial_len, t);
        else bad = use_rc ? nm_guard_list_one<true, true>(v, enc.data(), p, seq_len, ks, nk, t) : nm_guard_list_one<true, false>(v, enc.data(), p, seq_len, ks, nk, t);
        if (bad) return p;
    }
    return ~0ULL;
}

// k_period_runs + k_period_spread (nm_engine.hip): coarse[c] = coarse_stride for every stride of a tandem run whose first
// stride's walk of kmax + u - 1 bases still finds two occurrences, else 0.  periods (may be null): the period found per stride.
uint64_t hs_period_runs(hs_index *ix, const std::vector<nm_enc_word> &enc, uint64_t num_kmers, uint32_t kmax, uint32_t coarse_stride,
                        std::vector<uint32_t> &coarse, uint32_t *periods) {
    uint64_t steps = 0;
    const uint64_t n_coarse = (num_kmers + coarse_stride - 1) / coarse_stride;
    std::vector<uint32_t> raw(n_coarse, 0);
    coarse.assign(n_coarse, 0);
    const uint32_t len = coarse_stride + kmax - 1;
    for (uint64_t c = 0; c < n_coarse; c++) {
        const uint32_t u = nm_period_of(enc.data(), enc.size(), c * coarse_stride, len);
        if (periods) periods[c] = u;
        if (!u) continue;
        const bool first = c == 0 || nm_period_of(enc.data(), enc.size(), (c - 1) * coarse_stride, len) != u;
        if (!first) { raw[c] = NM_PERIOD_INHERIT; continue; }
        nm_tally t = {0, 0, 0, 0};
        uint32_t settled, exact;
        if (ix->big) nm_repeat_probe_ex<true>(ix->v, enc.data(), c * coarse_stride, kmax, u, t, settled, exact);
        else nm_repeat_probe_ex<false>(ix->v, enc.data(), c * coarse_stride, kmax, u, t, settled, exact);
        raw[c] = settled == u ? coarse_stride : 0u;
        steps += t.steps;
    }
    for (uint64_t c = 0; c < n_coarse; c++) {
        uint32_t v = raw[c];
        if (v == NM_PERIOD_INHERIT) {
            v = 0;
            for (uint64_t j = c; j-- > 0 && c - j <= 8192;)
                if (raw[j] != NM_PERIOD_INHERIT) { v = raw[j]; break; }
        }
        coarse[c] = v;
    }
    return steps;
}

// periodic != 0: the coarse words come from the tandem runs (k_period_runs) instead of the coarse probes
uint64_t hs_repeat_probes2(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmin,
                           uint32_t kmax, uint32_t stride, uint32_t coarse_stride, uint32_t *words, uint32_t *decided, int periodic,
                           uint32_t *periods, uint32_t *coarse_out);

uint64_t hs_repeat_probes(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmin,
                          uint32_t kmax, uint32_t stride, uint32_t coarse_stride, uint32_t *words, uint32_t *decided) {
    return hs_repeat_probes2(ix, seq, seq_len, num_kmers, kmin, kmax, stride, coarse_stride, words, decided, 0, nullptr, nullptr);
}

uint64_t hs_repeat_probes2(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmin,
                           uint32_t kmax, uint32_t stride, uint32_t coarse_stride, uint32_t *words, uint32_t *decided, int periodic,
                           uint32_t *periods, uint32_t *coarse_out) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    uint64_t steps = 0;
    const uint64_t n_probes = (num_kmers + stride - 1) / stride;
    // coarse probes first (k_repeat_probe_coarse; coarse_stride 0 = none): strides they settle completely get their
    // word without a walk (k_repeat_probe)
    std::vector<uint32_t> coarse;
    if (coarse_stride && periodic) {
        steps += hs_period_runs(ix, enc, num_kmers, kmax, coarse_stride, coarse, periods);
        if (coarse_out) for (size_t c = 0; c < coarse.size(); c++) coarse_out[c] = coarse[c];
    } else if (coarse_stride) {
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

// ---- the sites: k_sites -> gated repeat probes -> k_resolve (nm_engine.hip), block by block on the host --------
// Same helper functions (nm_core.h), same block geometry, same bitmaps; LDS arrays are vectors, lanes are loops.
// Needs the quad table (hs_check_quad builds it from the simulated seed table; core length = seed length).
// probes: 0 none, 1 fine, 2 coarse + fine.  list / n_list: list mode (kmin = first length, kmax = the longest).
// need_out (may be null): the bitmap k_sites leaves.  counters[0] = table entries read, [1] = positions walked,
// [2] = fine probes run, [3] = positions the probes decided, [4] = positions the second table settled.
// chance_max / walk_max: NM_SITE_CHANCE_MAX / NM_SITE_WALK_MAX of the device (256 / 64); tests also run other values.  Returns 0 ok, 8 k-mer not found, -1 not applicable.
int hs_sites(hs_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmin, uint32_t kmax, uint32_t d_cap,
             int probes, const uint32_t *list, uint32_t n_list, int elem_bytes, void *out, uint64_t *status,
             uint64_t *need_out, uint64_t *counters, uint32_t chance_max, uint32_t walk_max) {
    const nm_view &v = ix->v;
    const uint32_t m = v.quad_m;
    if (!v.quad || kmin < m + NM_QUAD_EXT || kmin > NM_SITE_MAX_KMIN || d_cap > NM_SITE_MAX_D) return -1;
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    const uint64_t n_enc_words = enc.size();
    for (int i = 0; i < 8; i++) status[i] = 0;
    status[2] = ~0ULL;
    for (int i = 0; i < 5; i++) counters[i] = 0;
    auto store = [&](uint64_t p, uint32_t val) {
        if (elem_bytes == 1) ((uint8_t *)out)[p] = (uint8_t)val;
        else if (elem_bytes == 2) ((uint16_t *)out)[p] = (uint16_t)val;
        else ((uint32_t *)out)[p] = val;
    };
    uint32_t d = kmin - (m + NM_QUAD_EXT);
    if (d > d_cap) d = d_cap;
    const uint32_t G = d + 5, BLOCK = 256, PER_LANE = 2, BP = BLOCK * PER_LANE * G, n_stage = NM_SITE_STAGE_WORDS(BP, kmax);
    const uint64_t n_need = (num_kmers + 63) / 64;
    std::vector<uint64_t> need(n_need + 1, 0);
    bool any_open = false;                                  // work[NM_WORK_OPEN]
    std::vector<nm_enc_word> enc_out(n_enc_words, nm_enc_word{~0ULL, ~0ULL, 0, ~0ULL});   // what k_sites leaves for the later kernels
    // ---- k_sites
    const uint64_t n_blocks = (num_kmers + BP - 1) / BP;
    for (uint64_t blk = 0; blk < n_blocks; blk++) {
        const uint64_t base = blk * BP, w0 = base >> 6;
        // phase 0: the block encodes its own stretch + lookahead from the raw bytes, 16 bytes per lane
        std::vector<nm_enc_word> s_enc(n_stage, nm_enc_word{0, 0, 0, 0});
        for (uint32_t t = 0; t < n_stage * 4; t++) {
            uint32_t lo, hi, amb;
            nm_encode_piece(seq, seq_len, (w0 + (t >> 2)) * 64 + (t & 3) * 16, (t & 1) != 0 && (((uintptr_t)seq) & 15u) == 0, lo, hi, amb);   // (both load forms)
            s_enc[t >> 2].lo |= (uint64_t)lo << (16 * (t & 3));
            s_enc[t >> 2].hi |= (uint64_t)hi << (16 * (t & 3));
            s_enc[t >> 2].amb |= (uint64_t)amb << (16 * (t & 3));
        }
        {
            const uint64_t own_end = w0 + BP / 64 < n_enc_words ? w0 + BP / 64 : n_enc_words;
            const uint64_t end = blk + 1 == n_blocks ? n_enc_words : own_end;
            for (uint64_t wi = w0; wi < end; wi++) enc_out[wi] = wi - w0 < n_stage ? s_enc[wi - w0] : nm_encode_word(seq, seq_len, wi, false);
        }
        std::vector<uint64_t> s_lo(n_stage), s_hi(n_stage), s_amb(n_stage);
        for (uint32_t i = 0; i < n_stage; i++) { s_lo[i] = s_enc[i].lo; s_hi[i] = s_enc[i].hi; s_amb[i] = s_enc[i].amb; }
        std::vector<uint32_t> s_set(BP / 32 + 2, 0), s_need(BP / 32, 0);
        auto lds_window = [&](uint32_t rel) {
            const uint32_t wi = rel >> 6, sh = rel & 63;
            nm_window w{s_lo[wi], s_hi[wi], s_amb[wi]};
            if (sh) {
                w.lo = (w.lo >> sh) | (s_lo[wi + 1] << (64 - sh));
                w.hi = (w.hi >> sh) | (s_hi[wi + 1] << (64 - sh));
                w.amb = (w.amb >> sh) | (s_amb[wi + 1] << (64 - sh));
            }
            return w;
        };
        for (uint32_t g = 0; g < BLOCK * PER_LANE; g++) {
            const nm_window win = lds_window(g * G + d);
            if (!(base + (uint64_t)g * G < num_kmers && nm_site_core_valid(win, m))) continue;
            counters[0]++;
            const uint64_t *entry = v.quad + nm_quad_slot(win, m) * NM_QUAD_WORDS;
            uint32_t b[4];
            nm_quad_index(win, m, b);
            const uint64_t *p01 = nm_quad_pair01(entry, b), *p34 = nm_quad_pair34(entry, b);
            const uint64_t e[4] = {p01[0], p01[1], p34[0], p34[1]};
            const uint64_t settled = nm_site_settled(nm_site_bits(win, m, b, e), d);
            if (!settled) continue;
            const uint32_t o = g * G, wi = o >> 5, sh = o & 31;
            s_set[wi] |= (uint32_t)(settled << sh);
            const uint64_t rest = sh ? settled >> (32 - sh) : settled >> 16 >> 16;
            if ((uint32_t)rest) s_set[wi + 1] |= (uint32_t)rest;
            if (rest >> 32) s_set[wi + 2] |= (uint32_t)(rest >> 32);
        }
        auto amb_word = [&](uint64_t i) -> uint64_t { return s_amb[i]; };
        const bool chance = v.quad2 != nullptr && kmin >= v.quad2_m + NM_QUAD_EXT;
        uint32_t open_total = 0;
        for (uint32_t j = 0; j < BP / 4; j++) {
            const uint32_t rel = 4 * j;
            const uint64_t q = base + rel;
            if (q >= num_kmers) break;
            const uint64_t left = num_kmers - q;
            const uint32_t inb = left >= 4 ? 0xFu : (1u << left) - 1u;
            uint32_t own_amb;
            const uint32_t valid = nm_valid4(amb_word, rel, kmin, own_amb) & inb;
            const uint32_t set4 = (s_set[rel >> 5] >> (rel & 31)) & 0xFu;
            uint32_t hit = valid & set4, open = valid & ~set4;
            status[0] += (uint64_t)__builtin_popcount(own_amb & inb);
            status[7] += (uint64_t)__builtin_popcount(~own_amb & inb);
            if (open) { s_need[rel >> 5] |= open << (rel & 31); open_total += (uint32_t)__builtin_popcount(open); }
            for (uint32_t t = 0; t < 4; t++)
                if ((inb >> t) & 1u) store(q + t, (hit >> t) & 1u ? kmin : 0u);
        }
        const bool use_dict = !list && v.dict && kmin >= v.dict_len;
        // second chance: the open positions of a block with few of them ask the table with the longer cores
        if (chance && open_total && open_total <= chance_max)
            for (uint32_t i = 0; i < BP / 32; i++)
                for (uint32_t bits = s_need[i]; bits; bits &= bits - 1) {
                    const uint32_t rel = i * 32 + (uint32_t)__builtin_ctz(bits);
                    counters[0]++;
                    if (nm_second_chance(v, lds_window(rel), kmin)) { store(base + rel, kmin); s_need[i] &= ~(1u << (rel & 31)); open_total--; counters[4]++; }
                }
        // repeat dictionary (k_sites phases 3D / 4D): a miss settles the position as kmin, a hit walks from the x-mer's interval
        bool dict_done = false;
        if (use_dict && open_total && open_total <= chance_max) {
            std::vector<std::pair<uint32_t, uint64_t>> walkers;
            for (uint32_t i = 0; i < BP / 32; i++)
                for (uint32_t bits = s_need[i]; bits; bits &= bits - 1) {
                    const uint32_t rel = i * 32 + (uint32_t)__builtin_ctz(bits);
                    counters[0]++;
                    uint64_t e;
                    if (nm_dict_find(v, nm_dict_key(lds_window(rel), v.dict_len), e)) walkers.push_back({rel, e});
                    else { store(base + rel, kmin); s_need[i] &= ~(1u << (rel & 31)); open_total--; counters[4]++; }
                }
            if (walkers.size() <= walk_max && kmax <= NM_SITE_LA_MAX) {
                for (auto &wk : walkers) {
                    const uint32_t rel = wk.first;
                    uint64_t lo, hi;
                    bool amb0 = false, err = false;
                    nm_tally t = {0, 0, 0, 0};
                    uint32_t val;
                    counters[1]++;
                    if (nm_seed_decode(wk.second, lo, hi))
                        val = ix->big ? nm_min_unique_walk<true, true>(v, s_enc.data(), rel, lds_window(rel), 0, lo, hi, v.dict_len, kmin, kmax, err, t)
                                      : nm_min_unique_walk<false, true>(v, s_enc.data(), rel, lds_window(rel), 0, lo, hi, v.dict_len, kmin, kmax, err, t);
                    else
                        val = ix->big ? nm_min_unique_one<true, true>(v, s_enc.data(), rel, kmin, kmax, amb0, err, t) : nm_min_unique_one<false, true>(v, s_enc.data(), rel, kmin, kmax, amb0, err, t);
                    if (err) { status[1] = 1; if (base + rel < status[2]) status[2] = base + rel; }
                    status[3] += t.steps; status[4] += t.blocks; status[6] += t.seeds;
                    store(base + rel, val);
                    s_need[rel >> 5] &= ~(1u << (rel & 31));
                }
                open_total = 0;
                dict_done = true;
            }
        }
        // a few open positions: the block finishes them itself; many: they stay for the probes and k_resolve
        const bool self = !dict_done && open_total && open_total <= walk_max && kmax <= NM_SITE_LA_MAX;
        if (self) {
            for (uint32_t i = 0; i < BP / 32; i++) {
                uint32_t bits = s_need[i];
                s_need[i] = 0;
                for (; bits; bits &= bits - 1) {
                    const uint64_t rel = i * 32 + (uint32_t)__builtin_ctz(bits), p = base + rel;
                    counters[1]++;
                    bool amb0 = false, err = false;
                    nm_tally t = {0, 0, 0, 0};
                    uint32_t val;                          // (on the block's staged words, positions relative to its first base)
                    if (list) val = ix->big ? nm_fixed_k_one<true, true>(v, s_enc.data(), rel, seq_len - base, list, n_list, amb0, err, t)
                                            : nm_fixed_k_one<false, true>(v, s_enc.data(), rel, seq_len - base, list, n_list, amb0, err, t);
                    else      val = ix->big ? nm_min_unique_one<true, true>(v, s_enc.data(), rel, kmin, kmax, amb0, err, t)
                                            : nm_min_unique_one<false, true>(v, s_enc.data(), rel, kmin, kmax, amb0, err, t);
                    if (err) { status[1] = 1; if (p < status[2]) status[2] = p; }
                    status[3] += t.steps; status[4] += t.blocks; status[6] += t.seeds;
                    store(p, val);
                }
            }
        } else if (open_total) any_open = true;
        for (uint32_t i = 0; i < BP / 64; i++)
            if (base + 64ull * i < num_kmers) need[w0 + i] = (uint64_t)s_need[2 * i] | ((uint64_t)s_need[2 * i + 1] << 32);
    }
    for (uint64_t i = 0; i < n_enc_words; i++)            // what k_sites left == the encode pass
        if (enc_out[i].lo != enc[i].lo || enc_out[i].hi != enc[i].hi || enc_out[i].amb != enc[i].amb) return -2;
    if (need_out) for (uint64_t i = 0; i < n_need; i++) need_out[i] = need[i];
    // ---- repeat probes where the bitmap is dense (k_repeat_probe_coarse, k_repeat_probe)
    std::vector<uint32_t> words;
    if (probes && any_open) {
        const uint32_t stride = 64;
        std::vector<uint32_t> coarse;
        if (probes == 2) {
            coarse.assign((num_kmers + NM_COARSE_STRIDE - 1) / NM_COARSE_STRIDE, 0);
            for (uint64_t c = 0; c < coarse.size(); c++) {
                const uint64_t j0 = c * (NM_COARSE_STRIDE / stride);
                if (!(j0 < n_need && nm_popc64(need[j0]) >= NM_PROBE_GATE_BITS)) continue;
                nm_tally t = {0, 0, 0, 0};
                uint32_t settled, exact;
                if (ix->big) nm_repeat_probe_ex<true>(v, enc.data(), c * NM_COARSE_STRIDE, kmax, NM_COARSE_STRIDE, t, settled, exact);
                else nm_repeat_probe_ex<false>(v, enc.data(), c * NM_COARSE_STRIDE, kmax, NM_COARSE_STRIDE, t, settled, exact);
                coarse[c] = settled;
            }
        }
        words.assign(n_need + 1, 0);
        for (uint64_t j = 0; j < n_need; j++) {
            const uint64_t P = j * stride;
            nm_tally t = {0, 0, 0, 0};
            if (!nm_probe_gate(need.data(), j, n_need)) continue;
            counters[2]++;
            if (probes == 2 && nm_coarse_covers(coarse[P / NM_COARSE_STRIDE], (uint32_t)(P % NM_COARSE_STRIDE), stride)) { words[j] = stride; continue; }
            words[j] = ix->big ? nm_repeat_probe<true>(v, enc.data(), P, kmax, stride, t) : nm_repeat_probe<false>(v, enc.data(), P, kmax, stride, t);
        }
    }
    // ---- k_sweep (NM_OPT_SWEEP): one lane per word, right to left, ONE extension per step (nm_core.h "the sweep")
    if (!any_open) probes = 0;
    if (g_sweep) {
        nm_sweep_args args;
        args.kmin = kmin; args.kmax = kmax; args.seq_len = seq_len; args.list = list; args.n_list = n_list; args.sc = v.superC;
        for (uint64_t cur = 0; any_open && cur < n_need; cur++) {
            uint64_t bits = need[cur];
            if (!bits) continue;
            if (probes) {
                const uint32_t wj = words[cur], wj1 = words[cur + 1];
                const uint32_t zeros = wj & 0xFFu;
                const uint64_t before = bits;
                bits &= zeros >= 64 ? 0ULL : ~((1ULL << zeros) - 1ULL);
                counters[3] += nm_popc64(before ^ bits);
                const uint32_t kj = wj >> 8, kj1 = wj1 >> 8;
                if (bits && kj1 && kj == kj1 + 64) {
                    for (; bits; bits &= bits - 1) {
                        const uint32_t o = (uint32_t)__builtin_ctzll(bits);
                        const uint32_t val = nm_sweep_element(enc.data(), args, cur * 64 + o, kj - o);
                        if (val) store(cur * 64 + o, val);
                        counters[3]++;
                    }
                }
            }
            nm_sweep st;
            nm_sweep_begin(st, cur, bits, enc[cur].lo, enc[cur].hi);
            nm_tally t = {0, 0, 0, 0};
            for (uint64_t guard = 0;; guard++) {
                if (guard > 64ull * ((uint64_t)kmax + 80)) return -3;     // (the state machine must end)
                uint64_t p = 0;
                uint32_t val = 0;
                const uint32_t ret = ix->big ? nm_sweep_step<true>(v, enc.data(), args, st, p, val, t) : nm_sweep_step<false>(v, enc.data(), args, st, p, val, t);
                if (ret & NM_SW_DONE) break;
                if (ret & NM_SW_ERR) {
                    bool amb0 = false, err = true;
                    if (list) val = ix->big ? nm_fixed_k_one<true, true>(v, enc.data(), p, seq_len, list, n_list, amb0, err, t)
                                            : nm_fixed_k_one<false, true>(v, enc.data(), p, seq_len, list, n_list, amb0, err, t);
                    if (err) { status[1] = 1; if (p < status[2]) status[2] = p; }
                }
                if (ret & NM_SW_EMIT) { store(p, val); counters[1]++; }
            }
            status[3] += t.steps; status[4] += t.blocks; status[6] += t.seeds;
        }
        any_open = false;
    }
    // ---- k_resolve
    for (uint64_t cur = 0; any_open && cur < n_need; cur++) {
        uint64_t bits = need[cur];
        uint32_t wj = 0, wj1 = 0;
        if (bits && probes) {
            wj = words[cur]; wj1 = words[cur + 1];
            const uint32_t zeros = wj & 0xFFu;
            const uint64_t before = bits;
            bits &= zeros >= 64 ? 0ULL : ~((1ULL << zeros) - 1ULL);
            counters[3] += nm_popc64(before ^ bits);
        }
        for (; bits; bits &= bits - 1) {
            const uint32_t o = (uint32_t)__builtin_ctzll(bits);
            const uint64_t p = cur * 64 + o;
            const uint32_t ks = probes ? nm_probe_kstar(wj, wj1, o, 64, kmax) : NM_PROBE_OPEN;
            if (ks != NM_PROBE_OPEN && (!list || ks > kmax)) {
                const uint32_t val = list ? 0u : nm_probe_element(ks, kmin, kmax, true);
                if (val) store(p, val);
                counters[3]++;
                continue;
            }
            counters[1]++;
            bool amb0 = false, err = false;
            nm_tally t = {0, 0, 0, 0};
            uint32_t val;
            if (list) val = ix->big ? nm_fixed_k_one<true, true>(v, enc.data(), p, seq_len, list, n_list, amb0, err, t)
                                    : nm_fixed_k_one<false, true>(v, enc.data(), p, seq_len, list, n_list, amb0, err, t);
            else      val = ix->big ? nm_min_unique_one<true, true>(v, enc.data(), p, kmin, kmax, amb0, err, t)
                                    : nm_min_unique_one<false, true>(v, enc.data(), p, kmin, kmax, amb0, err, t);
            if (err) { status[1] = 1; if (p < status[2]) status[2] = p; }
            status[3] += t.steps; status[4] += t.blocks; status[6] += t.seeds;
            store(p, val);
        }
    }
    status[5] += counters[0] * 4;
    return status[1] ? 8 : 0;
}

// nm_valid4 over a whole sequence: out[p] = 1 when the kmin bases from p on are free of ambiguity
void hs_valid_bits(const uint8_t *seq, uint64_t seq_len, uint32_t kmin, uint8_t *out) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    enc.resize(seq_len / 64 + 8, nm_enc_word{0, 0, ~0ULL, 0});
    auto amb_word = [&](uint64_t i) -> uint64_t { return enc[i].amb; };
    for (uint64_t q = 0; q < seq_len; q += 4) {
        uint32_t own;
        const uint32_t v = nm_valid4(amb_word, q, kmin, own);
        for (uint32_t t = 0; t < 4 && q + t < seq_len; t++) out[q + t] = (v >> t) & 1u;
    }
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

// ---- the FASTA scan of the native driver's parallel front-end (newmap_amd/csrc/nm_fasta_scan.hpp) ----
struct hs_fasta {
    const unsigned char *base = nullptr;
    size_t size = 0;
    std::vector<nm_fasta::Record> recs;
    unsigned threads = 1;
};

hs_fasta *hs_fasta_open(const char *path, unsigned threads, uint64_t chunk_bytes) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return nullptr;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return nullptr; }
    hs_fasta *h = new hs_fasta();
    h->size = (size_t)st.st_size;
    h->threads = threads;
    if (h->size) {
        void *m = mmap(nullptr, h->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); delete h; return nullptr; }
        h->base = (const unsigned char *)m;
    }
    close(fd);
    h->recs = nm_fasta::scan(h->base, h->size, threads, (size_t)chunk_bytes);
    return h;
}
void hs_fasta_close(hs_fasta *h) {
    if (!h) return;
    if (h->base) munmap((void *)h->base, h->size);
    delete h;
}
uint64_t hs_fasta_count(hs_fasta *h) { return h->recs.size(); }
// id (copied, NUL-terminated, at most cap - 1 bytes) and number of bases of record i; returns the id's length
uint64_t hs_fasta_record(hs_fasta *h, uint64_t i, char *id_out, uint64_t cap, uint64_t *n_bases) {
    const nm_fasta::Record &r = h->recs[i];
    const size_t n = r.id.size() < cap - 1 ? r.id.size() : (size_t)cap - 1;
    memcpy(id_out, r.id.data(), n);
    id_out[n] = 0;
    *n_bases = r.n_bases;
    return r.id.size();
}
// bases [lo, hi) of record i -> out (hi - lo bytes)
void hs_fasta_read(hs_fasta *h, uint64_t i, uint64_t lo, uint64_t hi, uint8_t *out) {
    std::vector<uint8_t> buf;
    const uint64_t b0 = nm_fasta::materialize(h->recs[i], lo, hi, h->threads, buf);
    memcpy(out, buf.data() + (lo - b0), (size_t)(hi - lo));
}

void hs_upper(const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmax, uint32_t *out) {
    std::vector<nm_enc_word> enc;
    hs_encode(seq, seq_len, enc);
    for (uint64_t p = 0; p < num_kmers; p++) out[p] = nm_upper_one(enc.data(), p, kmax);
}

}  // extern "C"

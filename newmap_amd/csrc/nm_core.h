// nm_core.h -- per-position search logic of the engine, written once.
//
// Included by nm_engine.hip with NM_HD = __device__ __forceinline__ (the product: every function
// here runs on the GPU only) and by the test-only host simulator tests/hostsim/hostsim.cpp with
// NM_HD = inline (lets the CPU test-suite exercise this exact logic against the oracle before
// a GPU is involved; that harness is not part of the product and is never loaded by it).
//
// What each function restates (file:line in the reference checkout):
//   nm_min_unique_one   newmap/search.py:383-548 binary_search, for ONE position, as the closed
//                       form of SURVEY.md Appendix A.2: least k in [kmin, U_p] whose total count
//                       is 1.  Because the total is >= 1 and non-increasing in k (:489-535 bisect
//                       a monotone predicate), the first k at which the suffix-array interval of
//                       the growing k-mer has size 1 decides the answer: max(k, kmin) if that
//                       still lies within U_p, else 0.
//   nm_fixed_k_one      newmap/search.py:551-644 linear_search for ONE position (list order,
//                       tail truncation :590, permanent drop on ambiguity :593-596, `count == 1`
//                       stop :639).
//   nm_count_fwd_one    src/newmap-count.c:91-206 count_kmers_from_sequence for ONE k-mer.
//   nm_upper_one        newmap/search.py:744-766 + :769-882 for ONE position.
//   the strand sum of newmap/search.py:647-697 is implicit: the index holds both strands
//   (nm_format.h), so one interval gives forward + reverse-complement occurrences; `--norc`
//   splits the interval with the strand blocks instead.
#ifndef NM_CORE_H
#define NM_CORE_H

#include <stdint.h>
#include "nm_format.h"
#ifndef NM_HASH_FN
#define NM_HASH_FN NM_HD
#endif
#include "nm_hash.h"

#ifndef NM_HD
#error "define NM_HD before including nm_core.h"
#endif

struct nm_enc_word {            // 64 sequence positions, produced by the encode kernel
    uint64_t lo, hi;            // bit-planes of the 2-bit base codes (A=0 C=1 G=2 T=3)
    uint64_t amb;               // 1 = byte not in ACGTacgt, or position >= seq_len
    uint64_t pad;
};

struct nm_view {                // the index as the kernels see it
    const nm_rank_block *rank;
    const nm_strand_block *strand;
    const uint64_t *sep;        // sorted BWT positions of separators
    const uint64_t *seed;       // 4^seed_len entries: interval start (40 bits) | size (24 bits, saturating)
    const uint64_t *superC;     // [n_super][4]: C[c] + occurrences of c before the superblock
    uint64_t n;                 // BWT length
    uint64_t n_sep;
    uint64_t C[4];              // == superC[0][*]
    uint32_t seed_len;
    uint32_t n_super;
    uint32_t seed_policy;       // experiment knob: cache policy of the seed-table load (0 = default)
    const nm_lf_entry *lfb;     // LF blocks: one 16-byte load per LF step (nullptr = use the packed rank blocks)
    const uint64_t *quad;       // quad table: 4^quad_m entries of 16 x u64 (nullptr = not built), see nm_quad_build_one
    uint32_t quad_m;            // its core length; it answers windows of quad_m + 4 bases
    const uint64_t *quad2;      // a second quad table with longer cores (nullptr = none): k_resolve's second chance
    uint32_t quad2_m;
    const uint64_t *hash_tab;   // NM_HASH_TAB_WORDS words (nm_hash.h): nibble tables + powers of the record fingerprint
    const nm_lf_entry *lf2;     // two-base LF blocks (below): 16 entries per 64 rows, nullptr = none
    const uint64_t *dict;       // repeat dictionary (below): 2^dict_bits buckets of 128 bytes, nullptr = none
    uint32_t dict_len;          // the length x of the strings it holds
    uint32_t dict_bits;
    const uint8_t *lcp;         // LCP bytes (nm_format.h: off_lcp), n + 1 of them, nullptr = the index file has none
};

struct nm_tally {               // counter build only
    uint32_t steps, blocks, seeds, strands;
};

#define NM_SEED_LO_BITS 40
#define NM_SEED_LO_MASK ((1ULL << NM_SEED_LO_BITS) - 1)
#define NM_SEED_CNT_SAT 0xFFFFFFu
#define NM_SEED_CNT_BITS_SHIFT NM_SEED_LO_BITS

// 0..3 for ACGTacgt, 4 otherwise.  Branch-free (a compare chain compiles to divergent branches per byte):
// with bit 5 cleared the four letters are 0x41 0x43 0x47 0x54 -- bits 1 and 2 spell A=00 C=01 G=11 T=10,
// and the letters are picked out of the 0x40..0x5F column by a 32-bit membership mask.
NM_HD uint32_t nm_base_code(uint32_t byte) {
    const uint32_t u = byte & 0xDFu;
    const uint32_t is_base = (uint32_t)((u >> 5) == 2u) & ((0x0010008Au >> (u & 31u)) & 1u);
    const uint32_t code = ((u >> 1) & 3u) ^ ((u >> 2) & 1u);
    return is_base ? code : 4u;
}

// Four sequence bytes at once (little-endian word: byte 0 is the first base): the 4-bit groups of the lo plane, the
// hi plane and the ambiguity plane.  Bit-sliced over the word -- bit K of every byte is brought to bit 7 of its byte
// by one shift (what spills into the neighbouring byte never reaches its bit 7), the letter test of nm_base_code
// becomes a dozen word-wide logic ops, and two shift-ors gather bits 7, 15, 23, 31 into the top nibble.
NM_HD void nm_base_codes4(uint32_t x, uint32_t &lo, uint32_t &hi, uint32_t &amb) {
    const uint32_t b0 = x << 7, b1 = x << 6, b2 = x << 5, b3 = x << 4, b4 = x << 3, b6 = x << 1, b7 = x;
    // with bit 5 ignored the letters are 010x0001 (A), 010x0011 (C), 010x0111 (G), 010x0100 with bit 4 set (T)
    const uint32_t frame = b6 & ~(b7 | b3);                        // bit 7 clear, bit 6 set, bit 3 clear
    const uint32_t acg = ~b4 & b0 & (b1 | ~b2);                    // bit 4 clear: ...001, ...011, ...111
    const uint32_t t = b4 & b2 & ~(b1 | b0);                       // bit 4 set:   ...100
    const uint32_t valid = frame & (acg | t) & 0x80808080u;
    auto top_nibble = [](uint32_t m) -> uint32_t {                // bits 7, 15, 23, 31 -> bits 0 .. 3
        m |= m << 7;                                              // 23 -> 30, 15 -> 22, 7 -> 14
        m |= m << 14;                                             // 15 -> 29, 14 -> 28
        return m >> 28;
    };
    lo = top_nibble((b1 ^ b2) & valid);                            // code bit 0 = bit 1 ^ bit 2, code bit 1 = bit 2
    hi = top_nibble(b2 & valid);
    amb = top_nibble(valid ^ 0x80808080u);
}

// 16 sequence bytes -> the 16-bit pieces of the three planes (k_encode16, and the encode stage of k_sites).  `off` is the
// byte offset of the piece in the segment; bytes at or past seq_len are ambiguous.  aligned16: seq + off is 16-byte aligned.
NM_HD void nm_encode_piece(const uint8_t *seq, uint64_t seq_len, uint64_t off, bool aligned16, uint32_t &lo, uint32_t &hi, uint32_t &amb) {
    uint32_t b[4] = {0, 0, 0, 0};
    if (off + 16 <= seq_len) {
        if (aligned16) {
            const uint32_t *p = (const uint32_t *)__builtin_assume_aligned(seq + off, 16);
            b[0] = p[0]; b[1] = p[1]; b[2] = p[2]; b[3] = p[3];
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 16; j++) b[j >> 2] |= (uint32_t)seq[off + j] << (8 * (j & 3));
        }
    } else {
        const uint32_t valid = off < seq_len ? (uint32_t)(seq_len - off) : 0;
        for (uint32_t j = 0; j < valid; j++) b[j >> 2] |= (uint32_t)seq[off + j] << (8 * (j & 3));
    }
    lo = hi = amb = 0;                                     // bytes past the end of the data are 0 in b[]: ambiguous
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        uint32_t l4, h4, a4;
        nm_base_codes4(b[j], l4, h4, a4);
        lo |= l4 << (4 * j);
        hi |= h4 << (4 * j);
        amb |= a4 << (4 * j);
    }
}

// one whole encoded word (64 bases from byte offset 64 * word) -- the few words a block of k_sites encodes alone
NM_HD nm_enc_word nm_encode_word(const uint8_t *seq, uint64_t seq_len, uint64_t word, bool aligned16) {
    nm_enc_word w = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t sub = 0; sub < 4; sub++) {
        uint32_t lo, hi, amb;
        nm_encode_piece(seq, seq_len, word * 64 + sub * 16, aligned16, lo, hi, amb);
        w.lo |= (uint64_t)lo << (16 * sub);
        w.hi |= (uint64_t)hi << (16 * sub);
        w.amb |= (uint64_t)amb << (16 * sub);
    }
    return w;
}

NM_HD uint32_t nm_popc64(uint64_t x) { return (uint32_t)__builtin_popcountll(x); }

// fingerprint term of word w of a segment (nm_hash.h): its bases at positions below `end` (the segment's num_kmers; a
// segment starts at a multiple of 64 of its record, so word w of the segment is word s / 64 + w of the record)
NM_HD uint64_t nm_hash_segment_word(const uint64_t *tab, const nm_enc_word &x, uint64_t w, uint64_t end) {
    const uint64_t p0 = w * 64;
    if (p0 >= end) return 0;
    const uint64_t take = end - p0 >= 64 ? ~0ULL : (1ULL << (end - p0)) - 1ULL;
    return nm_hash_word(x.lo & take, x.hi & take, x.amb & take) * nm_hash_word_power(tab, w);
}

// number of separator positions s with a <= s < b (b - a <= 64); only reached for the rare
// rank blocks flagged NM_SEP_FLAG
NM_HD uint32_t nm_sep_between(const nm_view &ix, uint64_t a, uint64_t b) {
    uint64_t lo = 0, hi = ix.n_sep;
    while (lo < hi) {                       // first separator >= a
        uint64_t mid = (lo + hi) >> 1;
        if (ix.sep[mid] < a) lo = mid + 1; else hi = mid;
    }
    uint32_t c = 0;
    while (lo < ix.n_sep && ix.sep[lo] < b) { c++; lo++; }
    return c;
}

// one 32-byte rank block held in registers
struct nm_blk { uint32_t c0, c1, c2, c3; uint64_t lo, hi; };

NM_HD nm_blk nm_load_blk(const nm_view &ix, uint64_t i) {
    const nm_rank_block *b = ix.rank + (i >> 6);
    nm_blk r;
    r.c0 = b->cnt[0]; r.c1 = b->cnt[1]; r.c2 = b->cnt[2]; r.c3 = b->cnt[3];
    r.lo = b->lo; r.hi = b->hi;
    return r;
}

// LF step on a loaded block: C[c] + rank_c(BWT, i)
template <bool BIG>
NM_HD uint64_t nm_lf_blk(const nm_view &ix, uint32_t c, uint64_t i, const nm_blk &b) {
    const uint32_t cc = c == 0 ? (b.c0 & ~NM_SEP_FLAG) : (c == 1 ? b.c1 : (c == 2 ? b.c2 : b.c3));
    uint64_t m = ((c & 1u) ? b.lo : ~b.lo) & ((c & 2u) ? b.hi : ~b.hi);
    const uint32_t off = (uint32_t)(i & 63);
    m &= (1ULL << off) - 1ULL;
    uint32_t r = cc + nm_popc64(m);
    if ((b.c0 & NM_SEP_FLAG) && c == 0 && off) r -= nm_sep_between(ix, i - off, i);
    if (BIG) return ix.superC[(i >> NM_SUPER_SHIFT) * 4 + c] + r;
    return (c == 0 ? ix.C[0] : (c == 1 ? ix.C[1] : (c == 2 ? ix.C[2] : ix.C[3]))) + r;
}

template <bool BIG>
NM_HD uint64_t nm_lf(const nm_view &ix, uint32_t c, uint64_t i) {
    if (ix.lfb) {                                         // one 16-byte load per step
        const nm_lf_entry *e = ix.lfb + ((i >> 6) * 4 + c);
        return e->base + nm_popc64(e->bits & ((1ULL << (i & 63)) - 1ULL));
    }
    return nm_lf_blk<BIG>(ix, c, i, nm_load_blk(ix, i));
}

// LF step of an interval [lo, hi): when both ends fall into the same 64-row block the second
// load is skipped (the common case once an interval is narrow)
template <bool BIG>
NM_HD void nm_lf_interval(const nm_view &ix, uint32_t c, uint64_t &lo, uint64_t &hi) {
    if (ix.lfb) {
        const nm_lf_entry *e = ix.lfb + ((lo >> 6) * 4 + c);
        const uint64_t base = e->base, bits = e->bits;
        const bool same = (lo >> 6) == (hi >> 6);
        uint64_t hbase = base, hbits = bits;
        if (!same) { const nm_lf_entry *f = ix.lfb + ((hi >> 6) * 4 + c); hbase = f->base; hbits = f->bits; }
        lo = base + nm_popc64(bits & ((1ULL << (lo & 63)) - 1ULL));
        hi = hbase + nm_popc64(hbits & ((1ULL << (hi & 63)) - 1ULL));
        return;
    }
    lo = nm_lf<BIG>(ix, c, lo);
    hi = nm_lf<BIG>(ix, c, hi);
}

// ---- two bases per step ------------------------------------------------------------------------------------------
// A walk is a chain of dependent loads, one per base; where walks are long (repeats: hundreds of bases) the chain IS the
// run time.  The two-base LF blocks halve it: prepending c1 and then c2 to a pattern maps row i to
//      C[c2] + rank_c2(C[c1]) + #{ rows i' < i : BWT[i'] = c1 and BWT[LF(i')] = c2 },
// i.e. a rank over the "BWT of the two preceding symbols" plus one of 16 constants -- per 64 rows and dinucleotide one
// 16-byte entry {constant + count before the block, indicator bits}: 256 bytes per 64 rows (4 n bytes: 25 GB for a
// 3 Gbp genome).  A walk takes two bases at a time while the interval keeps two or more rows (so the length at which it
// first holds one row is still found by single steps: the pair that would cross it is retried base by base).
// entry of block b, dinucleotide d = 4 * c2 + c1 at index 16 b + d.
template <bool BIG>
NM_HD void nm_lf2_interval(const nm_view &ix, uint32_t c1, uint32_t c2, uint64_t &lo, uint64_t &hi) {
    const uint32_t d = 4u * c2 + c1;
    const nm_lf_entry *e = ix.lf2 + ((lo >> 6) * 16 + d);
    const uint64_t base = e->base, bits = e->bits;
    uint64_t hbase = base, hbits = bits;
    if ((lo >> 6) != (hi >> 6)) { const nm_lf_entry *f = ix.lf2 + ((hi >> 6) * 16 + d); hbase = f->base; hbits = f->bits; }
    lo = base + nm_popc64(bits & ((1ULL << (lo & 63)) - 1ULL));
    hi = hbase + nm_popc64(hbits & ((1ULL << (hi & 63)) - 1ULL));
}

// the four LF entries of rank block b, from the packed structure (device build at open; host mirror)
template <bool BIG>
NM_HD void nm_lf_entries_of_block(const nm_view &ix, uint64_t b, nm_lf_entry out[4]) {
    const nm_rank_block *rb = ix.rank + b;
    uint64_t sepmask = 0;
    if (rb->cnt[0] & NM_SEP_FLAG) {
        uint64_t lo = 0, hi = ix.n_sep;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (ix.sep[mid] < b * 64) lo = mid + 1; else hi = mid; }
        for (; lo < ix.n_sep && ix.sep[lo] < b * 64 + 64; lo++) sepmask |= 1ULL << (ix.sep[lo] & 63);
    }
    const uint64_t rows = b * 64 < ix.n ? (ix.n - b * 64 >= 64 ? ~0ULL : ((1ULL << (ix.n - b * 64)) - 1ULL)) : 0ULL;
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        const uint32_t cnt = c == 0 ? (rb->cnt[0] & ~NM_SEP_FLAG) : rb->cnt[c];
        out[c].base = (BIG ? ix.superC[((b * 64) >> NM_SUPER_SHIFT) * 4 + c] : ix.C[c]) + cnt;
        out[c].bits = ((c & 1u) ? rb->lo : ~rb->lo) & ((c & 2u) ? rb->hi : ~rb->hi) & ~sepmask & rows;
    }
}

// suffixes of the RC half among suffix-array positions [0, i)
NM_HD uint64_t nm_strand_rank(const nm_view &ix, uint64_t i) {
    const nm_strand_block *b = ix.strand + (i >> 6);
    return b->before + nm_popc64(b->bits & ((1ULL << (i & 63)) - 1ULL));
}
NM_HD bool nm_strand_bit(const nm_view &ix, uint64_t i) {
    return (ix.strand[i >> 6].bits >> (i & 63)) & 1ULL;
}

// is BWT row i a base (not a separator), and which one
NM_HD bool nm_bwt_code(const nm_view &ix, uint64_t i, uint32_t &code) {
    const nm_rank_block *b = ix.rank + (i >> 6);
    const uint32_t off = (uint32_t)(i & 63);
    code = (uint32_t)((b->lo >> off) & 1ULL) | ((uint32_t)((b->hi >> off) & 1ULL) << 1);
    if (b->cnt[0] & NM_SEP_FLAG) return nm_sep_between(ix, i, i + 1) == 0;
    return true;
}

// the dinucleotide (4 * c2 + c1) in front of the suffix of row i, 16 = none (a separator among the two symbols)
template <bool BIG>
NM_HD uint32_t nm_bwt2_code(const nm_view &ix, uint64_t i) {
    uint32_t c1, c2;
    if (i >= ix.n || !nm_bwt_code(ix, i, c1)) return 16u;
    const uint64_t j = nm_lf<BIG>(ix, c1, i);
    if (!nm_bwt_code(ix, j, c2)) return 16u;
    return 4u * c2 + c1;
}

struct nm_window { uint64_t lo, hi, amb; };      // sequence positions [pos, pos+64)

NM_HD nm_window nm_window_from(const nm_enc_word &a, const nm_enc_word &b, uint32_t s) {
    nm_window r;
    r.lo = a.lo; r.hi = a.hi; r.amb = a.amb;
    if (s) {
        r.lo = (r.lo >> s) | (b.lo << (64 - s));
        r.hi = (r.hi >> s) | (b.hi << (64 - s));
        r.amb = (r.amb >> s) | (b.amb << (64 - s));
    }
    return r;
}

NM_HD nm_window nm_load_window(const nm_enc_word *enc, uint64_t pos) {
    const nm_enc_word *w = enc + (pos >> 6);
    return nm_window_from(w[0], w[1], (uint32_t)(pos & 63));
}

NM_HD uint32_t nm_window_code(const nm_window &w, uint32_t j) {
    return (uint32_t)((w.lo >> j) & 1ULL) | ((uint32_t)((w.hi >> j) & 1ULL) << 1);
}

// try to take the bases j and j + 1 of window w (both unambiguous) in one step: true = done (the interval keeps >= 2 rows).
// A failed try is a wasted load, and in unique sequence nearly every try would fail (the interval shrinks fourfold per
// base), so a walk only tries where the pair is likely to keep two rows: the interval is wide (>= NM_LF2_WIDE rows: a
// repeat family, or simply many rows left to shed), or `still` >= NM_LF2_STREAK single steps in a row left the row count
// unchanged -- the signature of a few exact copies; a failure clears `still`.
#ifndef NM_LF2_STREAK
#define NM_LF2_STREAK 2u
#endif
#ifndef NM_LF2_WIDE
#define NM_LF2_WIDE 16u
#endif
template <bool BIG>
NM_HD bool nm_lf2_try(const nm_view &ix, const nm_window &w, uint32_t j, uint64_t &lo, uint64_t &hi, uint32_t &still) {
    if ((still < NM_LF2_STREAK && hi - lo < NM_LF2_WIDE) || j >= 63 || ((w.amb >> j) & 3ULL)) return false;
    uint64_t l2 = lo, h2 = hi;
    nm_lf2_interval<BIG>(ix, 3u - nm_window_code(w, j), 3u - nm_window_code(w, j + 1), l2, h2);
    if (h2 < l2 + 2) { still = 0; return false; }
    lo = l2; hi = h2;
    return true;
}

// seed-table slot of the s-mer at the start of a window: low s bits = lo plane, next s = hi plane
NM_HD uint64_t nm_seed_slot(const nm_window &w, uint32_t s) {
    const uint64_t m = (1ULL << s) - 1ULL;
    return (w.lo & m) | ((w.hi & m) << s);
}
NM_HD uint32_t nm_seed_slot_code(uint64_t slot, uint32_t s, uint32_t j) {
    return (uint32_t)((slot >> j) & 1ULL) | ((uint32_t)((slot >> (s + j)) & 1ULL) << 1);
}

// are sequence positions [p+from, p+to) all unambiguous?  w/kbase: the window currently held
NM_HD bool nm_all_valid(const nm_enc_word *enc, uint64_t p, nm_window &w, uint32_t &kbase,
                        uint32_t from, uint32_t to) {
    uint32_t k = from;
    while (k < to) {
        uint32_t j = k - kbase;
        if (j >= 64) { w = nm_load_window(enc, p + k); kbase = k; j = 0; }
        const uint32_t span = (to - k) < (64 - j) ? (to - k) : (64 - j);
        const uint64_t mask = (span == 64 ? ~0ULL : ((1ULL << span) - 1ULL)) << j;
        if (w.amb & mask) return false;
        k += span;
    }
    return true;
}

// Range mode, second half: from an interval [lo, hi) that spells the first k bases of the k-mer at
// p (k = 0: the whole suffix array), extend base by base until the interval has one element.
// `w` holds sequence positions [p + kbase, p + kbase + 64).  Returns the element to store.
template <bool BIG, bool RC>
NM_HD uint32_t nm_min_unique_walk(const nm_view &ix, const nm_enc_word *enc, uint64_t p, nm_window w,
                                  uint32_t kbase, uint64_t lo, uint64_t hi, uint32_t k, uint32_t kmin,
                                  uint32_t kmax, bool &err, nm_tally &t) {
    uint32_t still = 0;                                   // single steps in a row that kept every row (nm_lf2_try)
    for (;;) {
        const uint64_t cnt = hi - lo;
        if (cnt == 0) { err = true; return 0; }           // search.py:699-722
        if (RC) {
            if (cnt == 1) break;
        } else {
            if (cnt == 1) {
                t.strands++;
                if (!nm_strand_bit(ix, lo)) { err = true; return 0; }
                break;
            }
            if (k >= kmin) {
                const uint64_t f = nm_strand_rank(ix, hi) - nm_strand_rank(ix, lo);
                t.strands += 2;
                if (f == 0) { err = true; return 0; }
                if (f == 1) return k;                     // least k >= kmin with forward count 1
            }
        }
        if (k >= kmax) return 0;
        uint32_t j = k - kbase;
        if (j >= 64) { w = nm_load_window(enc, p + k); kbase = k; j = 0; }
        if (RC && ix.lf2 && k + 2 <= kmax && nm_lf2_try<BIG>(ix, w, j, lo, hi, still)) {   // two bases, the interval still holds two rows
            t.steps += 2;
            t.blocks += 2;
            k += 2;
            continue;
        }
        if ((w.amb >> j) & 1ULL) return 0;                // k == U_p and still not unique
        const uint32_t c = 3u - nm_window_code(w, j);     // prepend the complement: search rc(k-mer)
        t.steps++;
        t.blocks += ((lo >> 6) == (hi >> 6)) ? 1u : 2u;
        nm_lf_interval<BIG>(ix, c, lo, hi);
        still = hi - lo == cnt ? still + 1u : 0u;
        k++;
    }
    const uint32_t ans = k > kmin ? k : kmin;             // k <= kmax here, kmin <= kmax
    if (!nm_all_valid(enc, p, w, kbase, k, ans)) return 0;   // ans must not exceed U_p
    return ans;
}

// Range mode, first half: what the window alone decides.  Returns true when the position is
// settled without touching the index (result 0): its own byte is ambiguous (amb0), or an ambiguous
// byte sits inside the first s <= kmin bases (U_p < kmin, search.py:437).
NM_HD bool nm_min_unique_settled(const nm_window &w, uint32_t s, bool use_seed, bool &amb0) {
    amb0 = (w.amb & 1ULL) != 0;
    if (amb0) return true;                                // search.py:399 mask -> finished, 0
    return use_seed && (w.amb & ((1ULL << s) - 1ULL)) != 0;
}

// the seed-table gather; NM_SEED_LOAD may be overridden by the device build to try cache policies
#ifndef NM_SEED_LOAD
#define NM_SEED_LOAD(ix, slot) ((ix).seed[(slot)])
#endif

// decode a seed entry into the interval of the first s bases; false = saturated entry (walk from 0)
NM_HD bool nm_seed_decode(uint64_t e, uint64_t &lo, uint64_t &hi) {
    const uint32_t c = (uint32_t)(e >> NM_SEED_LO_BITS);
    if (c == NM_SEED_CNT_SAT) return false;
    lo = e & NM_SEED_LO_MASK;
    hi = lo + c;
    return true;
}

// One position of range mode.  Returns the element to store (0 = nothing unique in range).
template <bool BIG, bool RC>
NM_HD uint32_t nm_min_unique_one(const nm_view &ix, const nm_enc_word *enc, uint64_t p,
                                 uint32_t kmin, uint32_t kmax, bool &amb0, bool &err, nm_tally &t) {
    const nm_window w = nm_load_window(enc, p);
    err = false;
    const uint32_t s = ix.seed_len;
    const bool use_seed = s && kmin >= s;
    if (nm_min_unique_settled(w, s, use_seed, amb0)) return 0;
    uint64_t lo = 0, hi = ix.n;
    uint32_t k = 0;
    if (use_seed) {
        t.seeds++;
        if (nm_seed_decode(NM_SEED_LOAD(ix, nm_seed_slot(w, s)), lo, hi)) k = s;
        else { lo = 0; hi = ix.n; }
    }
    return nm_min_unique_walk<BIG, RC>(ix, enc, p, w, 0, lo, hi, k, kmin, kmax, err, t);
}

// Repeat probe (both-strand range mode): one walk from every stride-th position P that knows two things
// the positions after it can use.
//  (1) A k-mer that is a substring of a string occurring twice occurs twice itself.  If S[P .. P+L) is still
//      not unique, every position q in [P, P + L - kmax] has its kmax-mer inside it: no unique prefix of any
//      length <= kmax, element 0 whatever kmin is (U_q = kmax: the span is free of ambiguous bytes).  If the
//      walk ends at an ambiguous byte (or the end of the data) with S[P .. P+k) not unique, every q in
//      [P, P+k) is 0: all its k-mers up to U_q = P + k - q lie inside the repeated string.
//  (2) e(q) = q + (least unique length at q) never decreases with q (same substring argument).  When the
//      probes of two neighbouring strides report the same end, e(P) = e(P + stride), every position between
//      them has that end too, so its least unique length is e - q -- exact, without a table lookup or a walk.
//      Inside any repeat longer than a stride that is the rule: all its positions become unique at the
//      first base after the repeat.
// The probe extends to at most kmax + stride - 1 bases.  It returns a word: bits 0..7 the number of positions
// from P on that (1) settles (0 .. stride), bits 8..31 the exact least unique length at P when the walk found
// it (0 = not known).  Never changes a result -- positions the probes do not decide take the ordinary path.
#define NM_PROBE_OPEN 0xFFFFFFFFu

// quad-table bit of the window that STARTS at the beginning of `w` (window 0 of the entry of the core four bases on,
// see the quad table below): true when that (quad_m + 4)-mer occurs exactly once
NM_HD bool nm_quad_once_first(const nm_view &ix, const nm_window &w, uint32_t kmax) {
    const uint32_t m = ix.quad_m, len = m + 4;
    if (len > kmax || (w.amb & ((1ULL << len) - 1ULL)) != 0) return false;
    const uint64_t mask = (1ULL << m) - 1ULL;
    const uint64_t slot = ((w.lo >> 4) & mask) | (((w.hi >> 4) & mask) << m);
    const uint32_t b0 = nm_window_code(w, 0) | (nm_window_code(w, 1) << 2) | (nm_window_code(w, 2) << 4) | (nm_window_code(w, 3) << 6);
    return ((ix.quad[slot * 16 + 2 * (b0 >> 6)] >> (b0 & 63u)) & 1ULL) != 0;
}

// (settled <= stride and exact are returned separately: the coarse probes use strides that do not fit the word)
template <bool BIG>
NM_HD void nm_repeat_probe_ex(const nm_view &ix, const nm_enc_word *enc, uint64_t P, uint32_t kmax,
                              uint32_t stride, nm_tally &t, uint32_t &settled, uint32_t &exact) {
    settled = 0;
    exact = 0;
    nm_window w;
    if (P & 63) {
        w = nm_load_window(enc, P);
    } else {                                              // probe positions are word-aligned: one 32-byte word, no shift
        const nm_enc_word &e0 = enc[P >> 6];
        w.lo = e0.lo; w.hi = e0.hi; w.amb = e0.amb;
    }
    if (w.amb & 1ULL) return;
    const uint32_t cap = kmax + stride - 1;
    const uint32_t s = ix.seed_len;
    uint64_t lo = 0, hi = ix.n;
    uint32_t k = 0, kbase = 0;
    if (ix.quad && nm_quad_once_first(ix, w, kmax)) return;     // unique within the quad table's window: nothing to tell
    if (ix.seed && s && s <= kmax && (w.amb & ((1ULL << s) - 1ULL)) == 0) {
        t.seeds++;
        if (nm_seed_decode(NM_SEED_LOAD(ix, nm_seed_slot(w, s)), lo, hi)) {
            if (hi - lo <= 1) return;                     // unique (or absent) within the seed: nothing to tell
            k = s;
        } else { lo = 0; hi = ix.n; }
    }
    uint32_t first_unique;                                // a lower bound of the least unique length at P
    uint32_t still = NM_LF2_STREAK;                       // (probes are only sent into dense repeats: try from the start)
    for (;;) {
        const uint64_t cnt = hi - lo;
        if (cnt == 0) return;                             // absent k-mer: the ordinary path reports it
        if (cnt == 1) { first_unique = exact = k; break; }   // every shorter prefix occurs twice
        if (k >= cap) { first_unique = cap + 1; break; }
        uint32_t j = k - kbase;
        if (j >= 64) { w = nm_load_window(enc, P + k); kbase = k; j = 0; }
        if (ix.lf2 && k + 2 <= cap && nm_lf2_try<BIG>(ix, w, j, lo, hi, still)) { t.steps += 2; t.blocks += 2; k += 2; continue; }
        if ((w.amb >> j) & 1ULL) {                        // S[P .. P+k) occurs twice and ends the run: all of it is 0
            settled = k < stride ? k : stride;
            return;
        }
        const uint32_t c = 3u - nm_window_code(w, j);
        t.steps++;
        t.blocks += ((lo >> 6) == (hi >> 6)) ? 1u : 2u;
        nm_lf_interval<BIG>(ix, c, lo, hi);
        still = hi - lo == cnt ? still + 1u : 0u;
        k++;
    }
    settled = first_unique > kmax ? first_unique - kmax : 0u;            // q - P < first_unique - kmax
    if (settled > stride) settled = stride;
    if (exact >= (1u << 24)) exact = 0;
}

template <bool BIG>
NM_HD uint32_t nm_repeat_probe(const nm_view &ix, const nm_enc_word *enc, uint64_t P, uint32_t kmax,
                               uint32_t stride, nm_tally &t) {
    uint32_t settled, exact;
    nm_repeat_probe_ex<BIG>(ix, enc, P, kmax, stride, t, settled, exact);
    return settled | (exact << 8);                         // stride <= 255
}

// Coarse probes: one walk per NM_COARSE_STRIDE positions, to at most kmax + NM_COARSE_STRIDE - 1 bases.  Deep inside
// a long repeat the eight fine probes of those positions would each walk kmax + 63 steps over nearly the same text;
// when the coarse probe settles a fine stride completely, its probe word is known (all 64 positions 0) without a walk.
#define NM_COARSE_STRIDE 512u
NM_HD bool nm_coarse_covers(uint32_t coarse_settled, uint32_t offset_in_coarse, uint32_t fine_stride) {
    return coarse_settled >= offset_in_coarse + fine_stride;
}

// ---- tandem repeats: a whole coarse stride settled by ONE walk per run -------------------------------------------
// If the text is u-periodic over a stretch -- S[i] == S[i + u], unambiguous, for every i in [P, P + len) -- every
// kmax-mer that starts in [P, P + len + u - kmax] is one of the u rotations of a single string.  A tandem array of
// period u that runs over many coarse strides is then ONE run of such stretches, and one walk of kmax + u - 1 bases from
// its first stride proves the whole run: if S[P .. P + kmax + u - 1) still occurs twice, each of the u rotations does,
// and every position of every stride of the run gets 0 (U_q = kmax: the stretch is free of ambiguous bytes).  The
// coarse probes would walk kmax + 511 bases for EACH stride of the run (nm_repeat_probe_ex).  The period is read off
// the encoded words alone; the index decides -- a sequence that is periodic but foreign to the index fails the walk
// and is searched the ordinary way.
#define NM_PERIOD_MAX 256u
#define NM_PERIOD_INHERIT 0x80000000u      /* coarse[] of a stride inside a run: take the value of the run's first stride */

// positions [64 w + u, 64 w + u + 64) of the planes, as a word
NM_HD nm_window nm_shifted_word(const nm_enc_word *enc, uint64_t w, uint32_t u) {
    const nm_enc_word &a = enc[w + (u >> 6)];
    return nm_window_from(a, enc[w + (u >> 6) + 1], u & 63u);
}

// least period u in [1, NM_PERIOD_MAX] of the stretch [P, P + len) in the sense above (P a multiple of 64), 0 = none.
// n_words: words of `enc` (positions past the data are ambiguous there, so a stretch that leaves the data fails).
NM_HD uint32_t nm_period_of(const nm_enc_word *enc, uint64_t n_words, uint64_t P, uint32_t len) {
    const uint64_t w0 = P >> 6;
    const uint32_t n_w = (len + 63u) >> 6;
    if (w0 + n_w + (NM_PERIOD_MAX >> 6) + 2 > n_words) return 0;
    const nm_enc_word first = enc[w0];
    const uint32_t head = len < 64u ? len : 64u;
    const uint64_t hmask = head == 64 ? ~0ULL : (1ULL << head) - 1ULL;
    if (first.amb & hmask) return 0;
    for (uint32_t j = 0; j <= (NM_PERIOD_MAX >> 6); j++) {
        const nm_enc_word a = enc[w0 + j], b = enc[w0 + j + 1];
        for (uint32_t sh = j ? 0u : 1u; sh < 64u && 64u * j + sh <= NM_PERIOD_MAX; sh++) {
            const nm_window y = nm_window_from(a, b, sh);
            if ((((first.lo ^ y.lo) | (first.hi ^ y.hi) | y.amb) & hmask) != 0) continue;
            // the first word agrees with itself u bases on: check the whole stretch
            const uint32_t u = 64u * j + sh;
            bool ok = true;
            for (uint32_t k = 1; k < n_w && ok; k++) {
                const nm_enc_word x = enc[w0 + k];
                const nm_window z = nm_shifted_word(enc, w0 + k, u);
                const uint32_t left = len - 64u * k;
                const uint64_t mk = left >= 64 ? ~0ULL : (1ULL << left) - 1ULL;
                ok = (((x.lo ^ z.lo) | (x.hi ^ z.hi) | x.amb | z.amb) & mk) == 0;
            }
            if (ok) return u;
        }
    }
    return 0;
}

// What the probe words of a position's stride (wj) and of the next stride (wj1) say about the position at
// offset o of the stride: its least unique length (kmax + 1 standing for "none up to kmax"), or NM_PROBE_OPEN.
NM_HD uint32_t nm_probe_kstar(uint32_t wj, uint32_t wj1, uint32_t o, uint32_t stride, uint32_t kmax) {
    if (o < (wj & 0xFFu)) return kmax + 1;
    const uint32_t kj = wj >> 8, kj1 = wj1 >> 8;
    if (kj1 && kj == kj1 + stride) return kj - o;          // same end on both sides
    return NM_PROBE_OPEN;
}

// element stored for a least unique length decided by the probes; kmin_valid: the first kmin bases of the
// position are unambiguous (only asked when kstar < kmin; the kstar bases themselves were walked over)
NM_HD uint32_t nm_probe_element(uint32_t kstar, uint32_t kmin, uint32_t kmax, bool kmin_valid) {
    if (kstar > kmax) return 0;
    if (kstar >= kmin) return kstar;
    return kmin_valid ? kmin : 0u;
}

// One position of list mode.
template <bool BIG, bool RC>
NM_HD uint32_t nm_fixed_k_one(const nm_view &ix, const nm_enc_word *enc, uint64_t p, uint64_t seq_len,
                              const uint32_t *ks, uint32_t nk, bool &amb0, bool &err, nm_tally &t) {
    nm_window w = nm_load_window(enc, p);
    uint32_t kbase = 0;
    amb0 = (w.amb & 1ULL) != 0;
    err = false;
    if (amb0) return 0;
    const uint64_t rem = seq_len - p;
    uint64_t lo = 0, hi = ix.n;
    uint32_t k = 0;
    uint32_t checked = 1;                                 // positions [p, p+checked) known valid
    uint32_t still = 0;
    const uint32_t s = ix.seed_len;
    for (uint32_t q = 0; q < nk; q++) {
        const uint32_t K = ks[q];
        const uint32_t L = (uint64_t)K < rem ? K : (uint32_t)rem;      // search.py:590 truncation
        if (L > checked) {
            if (!nm_all_valid(enc, p, w, kbase, checked, L)) return 0; // search.py:593-596 (permanent)
            checked = L;
        }
        if (L < k) { lo = 0; hi = ix.n; k = 0; }          // shorter than what was searched: restart
        if (k == 0 && s && L >= s) {
            if (kbase != 0) { w = nm_load_window(enc, p); kbase = 0; }
            const uint64_t e = ix.seed[nm_seed_slot(w, s)];
            t.seeds++;
            const uint32_t c = (uint32_t)(e >> NM_SEED_LO_BITS);
            if (c != NM_SEED_CNT_SAT) { lo = e & NM_SEED_LO_MASK; hi = lo + c; k = s; }
        }
        while (k < L && hi - lo > 1) {
            uint32_t j = k - kbase;
            if (j >= 64 || k < kbase) { w = nm_load_window(enc, p + k); kbase = k; j = 0; }
            if (RC && ix.lf2 && k + 2 <= L && nm_lf2_try<BIG>(ix, w, j, lo, hi, still)) { t.steps += 2; t.blocks += 2; k += 2; continue; }
            const uint32_t c = 3u - nm_window_code(w, j);
            const uint64_t before = hi - lo;
            t.steps++;
            t.blocks += ((lo >> 6) == (hi >> 6)) ? 1u : 2u;
            nm_lf_interval<BIG>(ix, c, lo, hi);
            still = hi - lo == before ? still + 1u : 0u;
            k++;
        }
        const uint64_t cnt = hi - lo;
        if (cnt == 0) { err = true; return 0; }
        bool uniq;
        if (RC) {
            uniq = cnt == 1;
        } else if (cnt == 1) {
            t.strands++;
            if (!nm_strand_bit(ix, lo)) { err = true; return 0; }
            uniq = true;
        } else {
            const uint64_t f = nm_strand_rank(ix, hi) - nm_strand_rank(ix, lo);
            t.strands += 2;
            if (f == 0) { err = true; return 0; }
            uniq = f == 1;
        }
        if (uniq) return K;                               // search.py:627-639
    }
    return 0;
}

// ---- several FASTA files in lock-step x several indexes (newmap/search.py:251-265, 461, 656-697) ----
// total(k) = sum over index f and sequence i of the count of sequence i's k-mer in index f (both
// strands, or forward only); a position is unique when total == number of sequences.  Mask, upper
// bound and record geometry come from the FIRST sequence (:263-265, :393-400).  Every (f, i) pair keeps
// its own interval; the total is non-increasing in k, so the first k with total <= n_seq decides.
#define NM_MAX_MULTI 4

struct nm_multi_args {
    nm_view view[NM_MAX_MULTI];
    const nm_enc_word *enc[NM_MAX_MULTI];
    uint32_t n_idx, n_seq;
};

struct nm_multi_state {
    uint64_t lo[NM_MAX_MULTI][NM_MAX_MULTI], hi[NM_MAX_MULTI][NM_MAX_MULTI];   // [index][sequence]
    nm_window w[NM_MAX_MULTI];
    uint32_t k, kbase;
};

NM_HD void nm_multi_reset(const nm_multi_args &a, nm_multi_state &st, uint64_t p) {
#pragma unroll
    for (uint32_t i = 0; i < NM_MAX_MULTI; i++)
        if (i < a.n_seq) st.w[i] = nm_load_window(a.enc[i], p);
#pragma unroll
    for (uint32_t f = 0; f < NM_MAX_MULTI; f++)
#pragma unroll
        for (uint32_t i = 0; i < NM_MAX_MULTI; i++) { st.lo[f][i] = 0; st.hi[f][i] = f < a.n_idx && i < a.n_seq ? a.view[f].n : 0; }
    st.k = 0;
    st.kbase = 0;
}

template <bool RC>
NM_HD uint64_t nm_multi_total(const nm_multi_args &a, const nm_multi_state &st) {
    uint64_t total = 0;
#pragma unroll
    for (uint32_t f = 0; f < NM_MAX_MULTI; f++)
#pragma unroll
        for (uint32_t i = 0; i < NM_MAX_MULTI; i++)
            if (f < a.n_idx && i < a.n_seq && st.hi[f][i] > st.lo[f][i])
                total += RC ? st.hi[f][i] - st.lo[f][i]
                            : nm_strand_rank(a.view[f], st.hi[f][i]) - nm_strand_rank(a.view[f], st.lo[f][i]);
    return total;
}

// extend every live interval by base k of its own sequence; false = the FIRST sequence is ambiguous there
NM_HD bool nm_multi_step(const nm_multi_args &a, nm_multi_state &st, uint64_t p) {
    uint32_t j = st.k - st.kbase;
    if (j >= 64) {
#pragma unroll
        for (uint32_t i = 0; i < NM_MAX_MULTI; i++)
            if (i < a.n_seq) st.w[i] = nm_load_window(a.enc[i], p + st.k);
        st.kbase = st.k;
        j = 0;
    }
    if ((st.w[0].amb >> j) & 1ULL) return false;
#pragma unroll
    for (uint32_t i = 0; i < NM_MAX_MULTI; i++) {
        if (i >= a.n_seq) continue;
        const bool amb_i = (st.w[i].amb >> j) & 1ULL;      // another sequence ambiguous here: it matches nothing
        const uint32_t c = 3u - nm_window_code(st.w[i], j);
#pragma unroll
        for (uint32_t f = 0; f < NM_MAX_MULTI; f++) {
            if (f >= a.n_idx || st.hi[f][i] <= st.lo[f][i]) continue;
            if (amb_i) { st.hi[f][i] = st.lo[f][i]; continue; }
            st.lo[f][i] = nm_lf<true>(a.view[f], c, st.lo[f][i]);
            st.hi[f][i] = nm_lf<true>(a.view[f], c, st.hi[f][i]);
        }
    }
    st.k++;
    return true;
}

template <bool RC>
NM_HD uint32_t nm_min_unique_multi_one(const nm_multi_args &a, uint64_t p, uint32_t kmin, uint32_t kmax,
                                       bool &amb0, bool &err) {
    nm_multi_state st;
    nm_multi_reset(a, st, p);
    amb0 = (st.w[0].amb & 1ULL) != 0;
    err = false;
    if (amb0) return 0;
    for (;;) {
        const uint64_t total = nm_multi_total<RC>(a, st);
        if (total == 0) { err = true; return 0; }             // search.py:699-722
        if (total <= a.n_seq) break;                          // == in the reference; < would never end there
        if (st.k >= kmax) return 0;
        if (!nm_multi_step(a, st, p)) return 0;
    }
    const uint32_t ans = st.k > kmin ? st.k : kmin;
    if (!nm_all_valid(a.enc[0], p, st.w[0], st.kbase, st.k, ans)) return 0;
    return ans;
}

template <bool RC>
NM_HD uint32_t nm_fixed_k_multi_one(const nm_multi_args &a, uint64_t p, uint64_t seq_len, const uint32_t *ks,
                                    uint32_t nk, bool &amb0, bool &err) {
    nm_multi_state st;
    nm_multi_reset(a, st, p);
    amb0 = (st.w[0].amb & 1ULL) != 0;
    err = false;
    if (amb0) return 0;
    const uint64_t rem = seq_len - p;
    uint32_t checked = 1;
    for (uint32_t q = 0; q < nk; q++) {
        const uint32_t K = ks[q];
        const uint32_t L = (uint64_t)K < rem ? K : (uint32_t)rem;
        if (L > checked) {
            nm_window w0 = st.w[0];
            uint32_t kb = st.kbase;
            if (!nm_all_valid(a.enc[0], p, w0, kb, checked, L)) return 0;
            checked = L;
        }
        if (L < st.k) nm_multi_reset(a, st, p);
        while (st.k < L)
            if (!nm_multi_step(a, st, p)) return 0;
        const uint64_t total = nm_multi_total<RC>(a, st);
        if (total == 0) { err = true; return 0; }
        if (total <= a.n_seq) return K;                       // search.py:627-636
    }
    return 0;
}

// forward-strand occurrences of one raw k-mer (any byte; non-ACGT -> 0)
template <bool BIG>
NM_HD uint32_t nm_count_fwd_one(const nm_view &ix, const uint8_t *kmer, uint64_t len, nm_tally &t) {
    uint64_t lo = 0, hi = ix.n;
    for (uint64_t j = 0; j < len; j++) {
        const uint32_t code = nm_base_code(kmer[j]);
        if (code > 3) return 0;
        const uint32_t c = 3u - code;
        t.steps++;
        lo = nm_lf<BIG>(ix, c, lo);
        hi = nm_lf<BIG>(ix, c, hi);
        if (lo >= hi) return 0;
    }
    if (len == 0) return 0;
    t.strands += 2;
    return (uint32_t)(nm_strand_rank(ix, hi) - nm_strand_rank(ix, lo));
}

NM_HD uint32_t nm_upper_one(const nm_enc_word *enc, uint64_t p, uint32_t kmax);

// forward-strand occurrences inside the interval [lo, hi) (both strands: its size)
template <bool RC>
NM_HD uint64_t nm_interval_count(const nm_view &ix, uint64_t lo, uint64_t hi) {
    if (hi <= lo) return 0;
    if (RC) return hi - lo;
    return nm_strand_rank(ix, hi) - nm_strand_rank(ix, lo);
}

// ---- the exact zero-count guard (newmap/search.py:699-722) for a record that is NOT one of the indexed records -------
// The reference raises when ANY k-mer its schedule asks the index about has a total of zero.  For a record whose
// fingerprint is in the index (nm_hash.h) that cannot happen and the fast paths run unchecked; every other record goes
// through these per-position functions after its search: they replay the reference's probe schedule for the position --
// the bisection of binary_search (search.py:424-433, 464-544) is a function of the least unique length alone -- and
// walk the LONGEST probed k-mer base by base: a prefix of a present string is present, so the reference raises for this
// position if and only if that walk runs empty.  ~100 LF steps per position instead of ~0.01: the price of a FASTA
// that is not the indexed genome, paid only then.

// the longest length the reference's bisection probes for a position with upper bound U and least unique length L
// (0 = no unique length up to U); initial_len = --initial-search-length (0 = none)
NM_HD uint32_t nm_ref_longest_probe(uint32_t kmin, uint32_t U, uint32_t L, uint32_t initial_len) {
    uint32_t lower = kmin, upper = U, q = (upper + lower) >> 1;          // :424-426
    if (initial_len && q > initial_len) q = initial_len;                  // :429-433
    uint32_t longest = 0;
    for (uint32_t it = 0; it < 64; it++) {
        if (q > longest) longest = q;
        if (L && q >= L) {                                                // total == 1
            if (q == lower || q == 0) break;                              // :504-508
            upper = q - 1;                                                // :524-527
        } else {                                                          // total > 1
            if (q == upper) break;                                        // :513-517
            lower = q + 1;                                                // :532-535
        }
        if (upper < lower && !(L && q >= L)) break;                       // (cannot happen for kmin <= U; keeps the loop finite)
        q = (upper + lower) >> 1;                                         // :540-542
    }
    return longest;
}

// range mode: would the reference raise for position p?  (U_p < kmin and ambiguous positions are never probed, :437)
template <bool BIG, bool RC>
NM_HD bool nm_guard_range_one(const nm_view &ix, const nm_enc_word *enc, uint64_t p, uint32_t kmin, uint32_t kmax,
                              uint32_t initial_len, nm_tally &t) {
    nm_window w = nm_load_window(enc, p);
    if (w.amb & 1ULL) return false;
    const uint32_t U = nm_upper_one(enc, p, kmax);
    if (U < kmin) return false;
    uint64_t lo = 0, hi = ix.n;
    uint32_t k = 0, kbase = 0, L = 0;
    while (k < U) {                                                       // least k with a total of one; an empty interval on the way: absent
        uint32_t j = k - kbase;
        if (j >= 64) { w = nm_load_window(enc, p + k); kbase = k; j = 0; }
        t.steps++;
        nm_lf_interval<BIG>(ix, 3u - nm_window_code(w, j), lo, hi);
        k++;
        const uint64_t c = nm_interval_count<RC>(ix, lo, hi);
        if (c == 0) return true;                                          // S[p .. p+k) is absent and k <= U: the bisection climbs to a probe >= k
        if (c == 1) { L = k; break; }
    }
    const uint32_t longest = nm_ref_longest_probe(kmin, U, L, initial_len);
    while (k < longest) {                                                 // (longest <= U: the window stays unambiguous)
        uint32_t j = k - kbase;
        if (j >= 64) { w = nm_load_window(enc, p + k); kbase = k; j = 0; }
        t.steps++;
        nm_lf_interval<BIG>(ix, 3u - nm_window_code(w, j), lo, hi);
        k++;
        if (nm_interval_count<RC>(ix, lo, hi) == 0) return true;
    }
    return false;
}

// list mode (search.py:551-644): every listed length is asked in turn until one has a total of one (:639) or the k-mer
// holds an ambiguous byte (:593-596, dropped for good); k-mers are cut at the end of the data (:590)
template <bool BIG, bool RC>
NM_HD bool nm_guard_list_one(const nm_view &ix, const nm_enc_word *enc, uint64_t p, uint64_t seq_len, const uint32_t *ks, uint32_t nk,
                             nm_tally &t) {
    nm_window w = nm_load_window(enc, p);
    if (w.amb & 1ULL) return false;
    const uint64_t rem = seq_len - p;
    uint64_t lo = 0, hi = ix.n;
    uint32_t k = 0, kbase = 0, checked = 1;
    for (uint32_t q = 0; q < nk; q++) {
        const uint32_t K = ks[q];
        const uint32_t L = (uint64_t)K < rem ? K : (uint32_t)rem;
        if (L > checked) {
            nm_window w2 = w;
            uint32_t kb2 = kbase;
            if (!nm_all_valid(enc, p, w2, kb2, checked, L)) return false;
            checked = L;
        }
        if (L < k) { lo = 0; hi = ix.n; k = 0; }
        while (k < L) {
            uint32_t j = k - kbase;
            if (j >= 64 || k < kbase) { w = nm_load_window(enc, p + k); kbase = k; j = 0; }
            t.steps++;
            nm_lf_interval<BIG>(ix, 3u - nm_window_code(w, j), lo, hi);
            k++;
            if (hi <= lo) return true;
        }
        const uint64_t c = nm_interval_count<RC>(ix, lo, hi);
        if (c == 0) return true;
        if (c == 1) return false;                                         // finished: nothing longer is asked
    }
    return false;
}

// per-position inclusive upper search length (ambiguous positions keep kmax)
NM_HD uint32_t nm_upper_one(const nm_enc_word *enc, uint64_t p, uint32_t kmax) {
    nm_window w = nm_load_window(enc, p);
    if (w.amb & 1ULL) return kmax;
    uint32_t k = 0, kbase = 0;
    while (k < kmax) {
        uint32_t j = k - kbase;
        if (j >= 64) { w = nm_load_window(enc, p + k); kbase = k; j = 0; }
        const uint64_t a = w.amb >> j;
        if (a) { k += (uint32_t)__builtin_ctzll(a); break; }
        k += 64 - j;
    }
    return k < kmax ? k : kmax;
}

// seed-table entry of slot `slot`: interval of the reverse complement of the s-mer it spells
template <bool BIG>
NM_HD uint64_t nm_seed_entry(const nm_view &ix, uint64_t slot, uint32_t s) {
    uint64_t lo = 0, hi = ix.n;
    for (uint32_t j = 0; j < s && lo < hi; j++) {
        const uint32_t c = 3u - nm_seed_slot_code(slot, s, j);
        lo = nm_lf<BIG>(ix, c, lo);
        hi = nm_lf<BIG>(ix, c, hi);
    }
    uint64_t cnt = hi > lo ? hi - lo : 0;
    if (cnt >= NM_SEED_CNT_SAT) cnt = NM_SEED_CNT_SAT;
    return cnt ? ((lo & NM_SEED_LO_MASK) | (cnt << NM_SEED_LO_BITS)) : 0;     // empty intervals are stored as 0
}

// ---- level-wise table construction: an s-mer's interval is ONE LF step away from the interval of
// its first s-1 bases, so level s is derived from level s-1 instead of walking s steps per entry
NM_HD uint64_t nm_seed_parent_slot(uint64_t slot, uint32_t s) {
    const uint64_t m = (1ULL << (s - 1)) - 1ULL;
    return (slot & m) | (((slot >> s) & m) << (s - 1));
}

template <bool BIG>
NM_HD uint64_t nm_seed_entry_from_parent(const nm_view &ix, uint64_t parent_entry, uint64_t slot, uint32_t s) {
    const uint32_t pc = (uint32_t)(parent_entry >> NM_SEED_LO_BITS);
    if (pc == NM_SEED_CNT_SAT) return nm_seed_entry<BIG>(ix, slot, s);      // parent size unknown: walk
    if (pc == 0) return 0;
    uint64_t lo = parent_entry & NM_SEED_LO_MASK, hi = lo + pc;
    const uint32_t c = 3u - nm_seed_slot_code(slot, s, s - 1);
    lo = nm_lf<BIG>(ix, c, lo);
    hi = nm_lf<BIG>(ix, c, hi);
    const uint64_t cnt = hi > lo ? hi - lo : 0;                              // < parent's, so not saturated
    return cnt ? ((lo & NM_SEED_LO_MASK) | (cnt << NM_SEED_LO_BITS)) : 0;
}

// ---- quad table: FOUR windows around one core in ONE 128-byte entry ---------------------------------
// Range mode only asks "which is the least length with one occurrence", and when a window of
// w = m + 4 bases already occurs once and w <= kmin the answer is kmin -- one BIT per w-mer is enough.
// The five windows that start at P .. P+4 share the m-mer core Y = S[P+4 .. P+4+m): the window at P+i is
// L.Y.R with the 4-i bases L before the core and the i bases R after it, 4^4 = 256 bits "L.Y.R occurs
// exactly once (both strands)" per window.  An entry holds the windows i = 0, 1, 3, 4 (the sites do not
// need i = 2, see below): 4 x 256 bits = one 128-byte line per core, nothing of the line is unused.
// Per window an 8-bit index b (bases in text order; the top two bits pick one of four 64-bit words):
//      i = 0:  l0 | l1<<2 | l2<<4 | l3<<6        i = 1:  l0 | l1<<2 | r0<<4 | l2<<6
//      i = 3:  r1 | r2<<2 | l0<<4 | r0<<6        i = 4:  r1 | r2<<2 | r3<<4 | r0<<6
// The word selector of windows 0 and 1 is the SAME base (the one just before the core), that of windows 3 and 4 too
// (the one just after it), and the words are laid out so that a lookup is TWO 16-byte loads from one line:
//      word 2 a + 0 / + 1      = windows 0 / 1 with selector a (base before the core)
//      word 8 + 2 c + 0 / + 1  = windows 3 / 4 with selector c (base after the core)
// (four scattered 8-byte loads of one line cost four address translations: on the 137 GB table the sites ran at
// half the line rate.)
// The table is derived from the seed table of length m without atomics: the lane of m-mer Z walks the
// (pruned) tree of its 256 four-base extensions; the leaves are the four words of window i = 4 of entry Z,
// and -- the index holds both strands, so a string and its reverse complement have the same count -- of
// window i = 0 of entry rc(Z).  The same leaves, read as Z[0] . (Z[1..m) b1) . b2 b3 b4, are a 16-bit piece of
// every word of window i = 3 of four entries and, mirrored, of window i = 1 of four more.  Every 16-bit piece
// of the table is written exactly once.
#define NM_QUAD_EXT 4u
#define NM_QUAD_WORDS 16u
#define NM_QUAD_OFFSETS 0x1Bu       /* window offsets an entry holds: bits 0, 1, 3, 4 */

NM_HD uint64_t nm_bit_reverse64(uint64_t x) {
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    return (x >> 32) | (x << 32);
}

// slot of the reverse complement of the m-mer a slot spells (reverse each plane, flip both)
NM_HD uint64_t nm_slot_revcomp(uint64_t slot, uint32_t m) {
    const uint64_t mask = (1ULL << m) - 1ULL;
    const uint64_t lo = nm_bit_reverse64(~(slot & mask)) >> (64 - m);
    const uint64_t hi = nm_bit_reverse64(~((slot >> m) & mask)) >> (64 - m);
    return (lo & mask) | ((hi & mask) << m);
}

// one extension step of the table walk: children of an interval; a one-row interval has exactly one
// child, named by its BWT symbol, which saves three of the four LF steps
template <bool BIG>
NM_HD void nm_quad_children(const nm_view &ix, uint64_t lo, uint64_t hi, uint64_t clo[4], uint64_t chi[4]) {
    for (uint32_t b = 0; b < 4; b++) { clo[b] = 0; chi[b] = 0; }
    if (hi <= lo) return;
    if (hi - lo == 1) {
        uint32_t c;
        if (!nm_bwt_code(ix, lo, c)) return;               // preceded by a separator: no extension
        const uint64_t r = nm_lf<BIG>(ix, c, lo);
        clo[3u - c] = r; chi[3u - c] = r + 1;              // appending base b = prepending its complement
        return;
    }
    for (uint32_t b = 0; b < 4; b++) {
        uint64_t l = lo, h = hi;
        nm_lf_interval<BIG>(ix, 3u - b, l, h);
        if (h > l) { clo[b] = l; chi[b] = h; }
    }
}

// all words of the quad table that the m-mer Z determines (ix.seed = seed table of length m)
template <bool BIG>
NM_HD void nm_quad_build_one(const nm_view &ix, uint64_t Z, uint32_t m, uint64_t *quad) {
    uint64_t lo = 0, hi = 0;
    if (!nm_seed_decode(ix.seed[Z], lo, hi)) {             // saturated size: walk the m bases
        lo = 0; hi = ix.n;
        for (uint32_t j = 0; j < m && lo < hi; j++) nm_lf_interval<BIG>(ix, 3u - nm_seed_slot_code(Z, m, j), lo, hi);
    }
    uint64_t w4[4] = {0, 0, 0, 0}, w0[4] = {0, 0, 0, 0};
    uint16_t p3[4][4] = {{0}}, p1[4][4] = {{0}};           // [b1][selector]: 16-bit pieces of windows 3 and 1
    uint64_t l1[4], h1[4], l2[4], h2[4], l3[4], h3[4], l4[4], h4[4];
    nm_quad_children<BIG>(ix, lo, hi, l1, h1);
    for (uint32_t b1 = 0; b1 < 4; b1++) {
        if (h1[b1] <= l1[b1]) continue;
        nm_quad_children<BIG>(ix, l1[b1], h1[b1], l2, h2);
        for (uint32_t b2 = 0; b2 < 4; b2++) {
            if (h2[b2] <= l2[b2]) continue;
            nm_quad_children<BIG>(ix, l2[b2], h2[b2], l3, h3);
            for (uint32_t b3 = 0; b3 < 4; b3++) {
                if (h3[b3] <= l3[b3]) continue;
                nm_quad_children<BIG>(ix, l3[b3], h3[b3], l4, h4);
                for (uint32_t b4 = 0; b4 < 4; b4++) {
                    if (h4[b4] - l4[b4] != 1) continue;    // Z b1 b2 b3 b4 occurs exactly once
                    const uint32_t c2 = 3u - b2, c3 = 3u - b3, c4 = 3u - b4;
                    w4[b1] |= 1ULL << (b2 | (b3 << 2) | (b4 << 4));                  // i = 4 of Z: r = b1 b2 b3 b4, selector r0 = b1
                    w0[3u - b1] |= 1ULL << (c4 | (c3 << 2) | (c2 << 4));             // i = 0 of rc(Z): l = ~b4 ~b3 ~b2 ~b1, selector l3
                    p3[b1][b2] |= (uint16_t)(1u << (b3 | (b4 << 2)));                // i = 3 of Z[1..m) b1: l0 = Z[0], r = b2 b3 b4, selector r0 = b2
                    p1[b1][c2] |= (uint16_t)(1u << (c4 | (c3 << 2)));                // i = 1 of its rc: l = ~b4 ~b3 ~b2, r0 = ~Z[0], selector l2 = ~b2
                }
            }
        }
    }
    const uint64_t mask = (1ULL << m) - 1ULL;
    const uint64_t zlo = Z & mask, zhi = (Z >> m) & mask;
    const uint32_t first = (uint32_t)(zlo & 1ULL) | ((uint32_t)(zhi & 1ULL) << 1);          // Z[0]
    const uint64_t rcz = nm_slot_revcomp(Z, m);
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        quad[Z * NM_QUAD_WORDS + 8 + 2 * j + 1] = w4[j];                                    // window 4, selector j
        quad[rcz * NM_QUAD_WORDS + 2 * j] = w0[j];                                          // window 0, selector j
    }
    for (uint32_t b1 = 0; b1 < 4; b1++) {
        const uint64_t core = (zlo >> 1) | ((uint64_t)(b1 & 1u) << (m - 1)) |
                              (((zhi >> 1) | ((uint64_t)(b1 >> 1) << (m - 1))) << m);      // Z[1..m) b1
        const uint64_t rcc = nm_slot_revcomp(core, m);
#pragma unroll
        for (uint32_t sel = 0; sel < 4; sel++) {
            ((uint16_t *)(quad + core * NM_QUAD_WORDS + 8 + 2 * sel))[first] = p3[b1][sel];       // window 3: piece l0 = Z[0]
            ((uint16_t *)(quad + rcc * NM_QUAD_WORDS + 2 * sel + 1))[3u - first] = p1[b1][sel];   // window 1: piece r0 = ~Z[0]
        }
    }
}

// core slot of the windows around the site whose first window starts `w`
NM_HD uint64_t nm_quad_slot(const nm_window &w, uint32_t m) {
    const uint64_t mask = (1ULL << m) - 1ULL;
    return ((w.lo >> NM_QUAD_EXT) & mask) | (((w.hi >> NM_QUAD_EXT) & mask) << m);
}

// bit indexes of the four windows (i = 0, 1, 3, 4) of the site whose first window starts `w`
NM_HD void nm_quad_index(const nm_window &w, uint32_t m, uint32_t b[4]) {
    const uint32_t c0 = nm_window_code(w, 0), c1 = nm_window_code(w, 1), c2 = nm_window_code(w, 2), c3 = nm_window_code(w, 3);
    const uint32_t r0 = nm_window_code(w, 4 + m), r1 = nm_window_code(w, 5 + m), r2 = nm_window_code(w, 6 + m), r3 = nm_window_code(w, 7 + m);
    b[0] = c0 | (c1 << 2) | (c2 << 4) | (c3 << 6);
    b[1] = c1 | (c2 << 2) | (r0 << 4) | (c3 << 6);
    b[2] = r1 | (r2 << 2) | (c3 << 4) | (r0 << 6);
    b[3] = r1 | (r2 << 2) | (r3 << 4) | (r0 << 6);
}

// the two 16-byte halves of a lookup: windows 0, 1 (selector b[0] >> 6 == b[1] >> 6) and windows 3, 4
NM_HD const uint64_t *nm_quad_pair01(const uint64_t *entry, const uint32_t b[4]) { return entry + 2 * (b[0] >> 6); }
NM_HD const uint64_t *nm_quad_pair34(const uint64_t *entry, const uint32_t b[4]) { return entry + 8 + 2 * (b[2] >> 6); }

// bit i of the result (i = 0, 1, 3, 4): the (m+4)-mer that starts i bases into the site occurs exactly once;
// e[0], e[1] = the pair at nm_quad_pair01, e[2], e[3] = the pair at nm_quad_pair34
NM_HD uint32_t nm_quad_bits(const uint32_t b[4], const uint64_t e[4]) {
    return (uint32_t)((e[0] >> (b[0] & 63u)) & 1ULL) | ((uint32_t)((e[1] >> (b[1] & 63u)) & 1ULL) << 1) |
           ((uint32_t)((e[2] >> (b[2] & 63u)) & 1ULL) << 3) | ((uint32_t)((e[3] >> (b[3] & 63u)) & 1ULL) << 4);
}

// ---- repeat dictionary: every x-mer that occurs MORE than once, with its interval ---------------------------------
// What the sites leave open is, on a genome without long repeats, a few percent of the positions whose windows happen to
// be repeated; nearly all of them are unique a base or two further on.  The dictionary answers that with ONE 128-byte line:
// it holds every x-mer of the both-strand text that occurs at least twice (x = ceil(log4 n) + 3: 20 for a 3 Gbp genome --
// 17 M of the 1.1 T possible 20-mers) with its suffix-array interval, in buckets of 8 entries {key, interval}.  An open
// position looks its x-mer up (x <= kmin): a MISS means the x-mer occurs once, so the element is kmin; a HIT hands the
// walk an interval x bases deep.  This replaces the second quad table's line, the seed entry and the first x - 16 LF
// steps of a walk.  (An x-mer that is ABSENT from the index misses too: that is the guard's business, nm_hash.h.)
// key = the x-mer's bit-planes (lo | hi << 32, x <= 24); value = interval start (40 bits) | size (24 bits, saturating).
#define NM_DICT_SLOTS 8u
#define NM_DICT_EMPTY (~0ULL)
#define NM_DICT_MAX_LEN 24u
#define NM_DICT_MAX_PROBES 16u

NM_HD uint64_t nm_dict_key(const nm_window &w, uint32_t x) {
    const uint64_t m = (1ULL << x) - 1ULL;
    return (w.lo & m) | ((w.hi & m) << 32);
}
NM_HD uint64_t nm_dict_bucket(uint64_t key, uint32_t bits) {
    uint64_t z = key * 0x9E3779B97F4A7C15ULL;
    z ^= z >> 29;
    z *= 0xBF58476D1CE4E5B9ULL;
    return z >> (64 - bits);
}
// one-lane lookup (walk kernels, host mirror): true = found, value in `entry`
NM_HD bool nm_dict_find(const nm_view &ix, uint64_t key, uint64_t &entry) {
    uint64_t b = nm_dict_bucket(key, ix.dict_bits);
    const uint64_t mask = (1ULL << ix.dict_bits) - 1ULL;
    for (uint32_t probe = 0; probe < NM_DICT_MAX_PROBES; probe++, b = (b + 1) & mask) {
        const uint64_t *e = ix.dict + b * (2 * NM_DICT_SLOTS);
        for (uint32_t i = 0; i < NM_DICT_SLOTS; i++) {
            const uint64_t k = e[2 * i];
            if (k == key) { entry = e[2 * i + 1]; return true; }
            if (k == NM_DICT_EMPTY) return false;          // (a bucket fills from its first slot: nothing beyond an empty one)
        }
    }
    return false;
}
// children of a node {planes klo / khi of an L-mer, its seed-format entry} that occur at least twice: appended base b, its
// planes and entry.  Returns the number written (0 .. 4).
template <bool BIG>
NM_HD uint32_t nm_dict_children(const nm_view &ix, uint32_t klo, uint32_t khi, uint64_t entry, uint32_t L, uint32_t out_lo[4], uint32_t out_hi[4],
                                uint64_t out_entry[4]) {
    uint64_t lo, hi;
    if (!nm_seed_decode(entry, lo, hi)) {                  // saturated size: walk the L bases
        lo = 0; hi = ix.n;
        for (uint32_t j = 0; j < L && lo < hi; j++) {
            const uint32_t code = ((klo >> j) & 1u) | (((khi >> j) & 1u) << 1);
            nm_lf_interval<BIG>(ix, 3u - code, lo, hi);
        }
    }
    if (hi - lo < 2) return 0;
    uint32_t n = 0;
    for (uint32_t b = 0; b < 4; b++) {
        uint64_t l = lo, h = hi;
        nm_lf_interval<BIG>(ix, 3u - b, l, h);
        if (h <= l || h - l < 2) continue;
        uint64_t cnt = h - l;
        if (cnt >= NM_SEED_CNT_SAT) cnt = NM_SEED_CNT_SAT;
        out_lo[n] = klo | ((b & 1u) << L);
        out_hi[n] = khi | ((b >> 1) << L);
        out_entry[n] = (l & NM_SEED_LO_MASK) | (cnt << NM_SEED_CNT_BITS_SHIFT);
        n++;
    }
    return n;
}

// ---- sites: one quad entry settles 5 + d positions (nm_engine.hip: k_sites, k_resolve) -----------
// A string that contains a string occurring once occurs once itself.  If the w-mer (w = m + 4) that starts at P + i
// occurs exactly once, then every position q with  q <= P + i  and  P + i + w <= q + kmin  has a kmin-mer that
// contains it: its least unique length is <= kmin and its element is kmin (given kmin unambiguous bases).  With
// d = kmin - w that is q in [P + i - d, P + i].  A SITE is the entry lookup at P = p0 + d for the GROUP of
// G = d + 5 positions p0 .. p0 + d + 4: window i of the entry (i = 0, 1, 3, 4) settles the positions t = q - p0 in
// [i, i + d], so for d >= 1 the four windows together cover the whole group (for d = 0 position 2 stays open) and the
// table is read once per G positions.  What no window settles (it is repeated, or it holds an ambiguous base) is
// left open: second table, repeat probes, then the seed table and the walk.  The reference asks the index about
// every position on its own (newmap/search.py:383-548); results are identical.
#define NM_SITE_MAX_D 59u           /* G <= 64: a group's settled bits fit one word */
#define NM_SITE_MAX_KMIN 252u       /* the kmin bases of a block's last position lie inside the words the block stages */
#define NM_SITE_LA_MAX 448u         /* kmax up to here: a block stages the lookahead of its own walks too (else they go to k_resolve) */
// words a block of BP positions stages: its own, the lookahead of the validity test (kmin <= 252) and -- for
// kmax <= NM_SITE_LA_MAX -- of the walks it finishes itself (a walk reads the two words at p + k, k < kmax)
#define NM_SITE_STAGE_WORDS(bp, kmax) ((bp) / 64 + ((kmax) <= NM_SITE_LA_MAX && ((kmax) + 64) / 64 + 2 > 5 ? ((kmax) + 64) / 64 + 2 : 5))

// the core of the site whose 64-base window is `w` lies in unambiguous bases
NM_HD bool nm_site_core_valid(const nm_window &w, uint32_t m) {
    return ((w.amb >> NM_QUAD_EXT) & ((1ULL << m) - 1ULL)) == 0;
}

// bit i (i = 0, 1, 3, 4): the (m+4)-mer i bases into the site occurs exactly once AND its window is free of ambiguity
NM_HD uint32_t nm_site_bits(const nm_window &w, uint32_t m, const uint32_t b[4], const uint64_t e[4]) {
    const uint32_t once = nm_quad_bits(b, e);
    const uint64_t wm = (1ULL << (m + NM_QUAD_EXT)) - 1ULL;
    uint32_t ok = 0;
#pragma unroll
    for (uint32_t i = 0; i < 5; i++) ok |= (uint32_t)(((w.amb >> i) & wm) == 0) << i;
    return once & ok & NM_QUAD_OFFSETS;
}

// bit t of the result: position t of the group (t = 0 .. d + 4) is settled by one of the site's windows
NM_HD uint64_t nm_site_settled(uint32_t bits5, uint32_t d) {
    const uint64_t run = (1ULL << (d + 1)) - 1ULL;        // d <= NM_SITE_MAX_D
    uint64_t s = 0;
#pragma unroll
    for (uint32_t i = 0; i < 5; i++)
        if ((bits5 >> i) & 1u) s |= run << i;
    return s;
}

// bit t of the result (t = 0 .. 3): the kmin bases from position q + t on are free of ambiguity (U_p >= kmin,
// newmap/search.py:437).  q is a multiple of 4; amb_word(i) = ambiguity plane of bases [64 i, 64 i + 64) in q's
// coordinates; words up to (q + 3 + kmin) / 64 + 1 are read.
template <class AmbWord>
NM_HD uint32_t nm_valid4(AmbWord amb_word, uint64_t q, uint32_t kmin, uint32_t &own_amb4) {
    const uint64_t wi = q >> 6;
    const uint32_t sh = (uint32_t)(q & 63);
    auto window = [&](uint64_t j) -> uint64_t {             // bases q + 64 j .. q + 64 j + 63
        const uint64_t a = amb_word(wi + j);
        return sh ? (a >> sh) | (amb_word(wi + j + 1) << (64 - sh)) : a;
    };
    const uint64_t A = window(0);
    own_amb4 = (uint32_t)(A & 0xFu);
    // first ambiguous base at or after q + 64, as an offset from q (looked for up to q + kmin + 2)
    uint64_t first = ~0ULL;
    for (uint32_t off = 64; off < kmin + 3; off += 64) {
        const uint64_t b = window(off >> 6);
        if (b) { first = off + (uint32_t)__builtin_ctzll(b); break; }
    }
    uint32_t valid = 0;
#pragma unroll
    for (uint32_t t = 0; t < 4; t++) {
        const uint32_t n_head = kmin < 64u - t ? kmin : 64u - t;           // bases of the kmin-mer inside A
        const uint64_t mask = n_head == 64 ? ~0ULL : (1ULL << n_head) - 1ULL;
        bool ok = ((A >> t) & mask) == 0;
        if (kmin > 64u - t) ok = ok && first >= (uint64_t)t + kmin;
        valid |= (uint32_t)ok << t;
    }
    return valid;
}

// Second chance: the sites of a launch may read a table with SHORT cores -- large groups, few table lines, but more
// windows that are repeated.  A position they leave open first asks the table with longer cores: the entry at P = p
// holds the windows at p + 0, 1, 3, 4, and any of them that occurs once inside the position's kmin-mer (i <= kmin - w2)
// settles it.  One line instead of the seed entry plus the rank lines of a walk.  `w` = the 64 bases from p on.
// (the same with the entry's words already fetched: b = nm_quad_index(w, quad2_m), e = both pairs of the entry)
NM_HD bool nm_second_chance_bits(const nm_view &ix, const nm_window &w, uint32_t kmin, const uint32_t b[4], const uint64_t e[4]) {
    const uint32_t m = ix.quad2_m, reach = kmin - (m + NM_QUAD_EXT);
    const uint32_t usable = reach >= 4 ? 0x1Fu : (1u << (reach + 1)) - 1u;
    return (nm_site_bits(w, m, b, e) & usable) != 0;
}
NM_HD bool nm_second_chance(const nm_view &ix, const nm_window &w, uint32_t kmin) {
    const uint32_t m = ix.quad2_m, len = m + NM_QUAD_EXT;
    if (!nm_site_core_valid(w, m)) return false;
    const uint64_t *entry = ix.quad2 + nm_quad_slot(w, m) * NM_QUAD_WORDS;
    uint32_t b[4];
    nm_quad_index(w, m, b);
    const uint32_t reach = kmin - len;                     // windows p + i with i <= reach lie inside the kmin-mer
    const uint32_t usable = reach >= 4 ? 0x1Fu : (1u << (reach + 1)) - 1u;
    const uint64_t *p01 = nm_quad_pair01(entry, b), *p34 = nm_quad_pair34(entry, b);
    const uint64_t e[4] = {p01[0], p01[1], usable & 0x18u ? p34[0] : 0ULL, usable & 0x18u ? p34[1] : 0ULL};
    return (nm_site_bits(w, m, b, e) & usable) != 0;
}

// ---- both directions: the rows of a string AND of its reverse complement ----------------------------------------------
// The text holds every run together with its reverse complement (nm_format.h), so a string X and rc(X) occur equally
// often, and the separators sort before A.  Let [k, k + s) be the rows of X and [l, l + s) those of rc(X).  Prepending c
// to X is one LF step at both ends of [k, k + s); it APPENDS comp(c) to rc(X), whose rows inside [l, l + s) follow those of
// rc(X) before a separator (as many as rows of X after one: s minus the four extension sizes) and of rc(X) y for
// y < comp(c) (as many as rows of x X for x > c).  So one step, in either direction, reads the rank blocks at the two ends
// of ONE of the intervals and moves both (the FMD-index of Li 2012, restated for this text and block layout).
struct nm_bi { uint64_t k, l, s; };

// occurrences of the four bases in BWT[superblock start .. i) from one loaded block (three popcounts: T = both planes,
// C and G = one plane each minus T, A = the rest minus the separators, which are stored as A)
NM_HD void nm_rank4_blk(const nm_view &ix, uint64_t i, const nm_blk &b, uint32_t r[4]) {
    const uint32_t off = (uint32_t)(i & 63);
    const uint64_t below = (1ULL << off) - 1ULL;
    const uint64_t lo = b.lo & below, hi = b.hi & below;
    const uint32_t n_lo = nm_popc64(lo), n_hi = nm_popc64(hi), n_t = nm_popc64(lo & hi);
    uint32_t sep = 0;
    if ((b.c0 & NM_SEP_FLAG) && off) sep = nm_sep_between(ix, i - off, i);
    r[0] = (b.c0 & ~NM_SEP_FLAG) + (off - (n_lo + n_hi - n_t)) - sep;
    r[1] = b.c1 + (n_lo - n_t);
    r[2] = b.c2 + (n_hi - n_t);
    r[3] = b.c3 + n_t;
}

// x = {rows of X, rows of rc(X), size} -> the same for c X, from the rank blocks of x.k (ba) and x.k + x.s (bb) and the four
// C[c] + counts before the superblock of x.k (cs)
template <bool BIG>
NM_HD void nm_bi_extend_blk(const nm_view &ix, nm_bi &x, uint32_t c, const nm_blk &ba, const nm_blk &bb, const uint64_t cs[4]) {
    uint32_t a[4], b[4];
    const uint64_t end = x.k + x.s;
    nm_rank4_blk(ix, x.k, ba, a);
    nm_rank4_blk(ix, end, bb, b);
    uint64_t s0 = (uint64_t)b[0] - a[0], s1 = (uint64_t)b[1] - a[1], s2 = (uint64_t)b[2] - a[2], s3 = (uint64_t)b[3] - a[3];
    if (BIG && (end >> NM_SUPER_SHIFT) != (x.k >> NM_SUPER_SHIFT)) {      // the interval crosses a superblock start: counts relative to two of them
        const uint64_t *cb = ix.superC + (end >> NM_SUPER_SHIFT) * 4;
        s0 += cb[0] - cs[0]; s1 += cb[1] - cs[1]; s2 += cb[2] - cs[2]; s3 += cb[3] - cs[3];
    }
    const uint64_t after_sep = x.s - (s0 + s1 + s2 + s3);
    const uint64_t above = c == 0 ? s1 + s2 + s3 : (c == 1 ? s2 + s3 : (c == 2 ? s3 : 0ULL));
    x.l += after_sep + above;
    x.k = c == 0 ? cs[0] + a[0] : (c == 1 ? cs[1] + a[1] : (c == 2 ? cs[2] + a[2] : cs[3] + a[3]));
    x.s = c == 0 ? s0 : (c == 1 ? s1 : (c == 2 ? s2 : s3));
}

// (the same, loading what it needs: table construction checks, tests)
template <bool BIG>
NM_HD void nm_bi_extend(const nm_view &ix, nm_bi &x, uint32_t c, nm_tally &t) {
    const uint64_t end = x.k + x.s;
    const bool same = (x.k >> 6) == (end >> 6);
    const nm_blk ba = nm_load_blk(ix, x.k);
    const uint64_t *cs = BIG ? ix.superC + (x.k >> NM_SUPER_SHIFT) * 4 : ix.C;
    const uint64_t c4[4] = {cs[0], cs[1], cs[2], cs[3]};
    t.steps++;
    t.blocks += same ? 1u : 2u;
    nm_bi_extend_blk<BIG>(ix, x, c, ba, same ? ba : nm_load_blk(ix, end), c4);
}

// ---- the sweep: consecutive open positions share their walks (k_sweep) ------------------------------------------------
// e(q) = q + (least unique length at q) never decreases with q.  Take the open positions of a stretch from RIGHT to LEFT
// and keep, for the position q just decided, the rows of a string S[q .. F) that still occurs twice -- with F + 1 = e(q)
// when the walk found it (EXACT), or F = the end of what a walk may read (kmax bases, an ambiguous byte).  ONE step to the
// left gives the rows of S[q - 1 .. F).  Two rows or more: S[q - 1 .. F + 1) contains the string that occurs once and
// S[q - 1 .. F) does not occur once, so e(q - 1) = e(q): the position is decided by that one step (not EXACT: nothing
// up to F occurs once, element 0 as for q).  One row: the end moved left.  Where to is read off the index's LCP bytes
// (nm_format.h; mode NM_SW_LCP below): the row that is left holds the only occurrence of S[q - 1 .. F), the larger of its two
// bytes is the longest prefix its suffix shares with any other, one base more is the least unique length, and the rows whose
// bytes reach it carry the chain on.  Without the bytes (an index file written before them, a handle of the one-shot CLI, a
// capped byte with kmax > 255) the position walks for itself -- seed table, then base by base to the right, BOTH intervals
// kept, because the chain afterwards grows the string to the left -- as the first open position of every word does.
// Where a stretch of a repeat family member differs from its nearest relative every 5 - 20 bases, a position costs one step
// and a moved end one more read -- the reference pays ~7 probes of 20 - 200 bases for every position on its own
// (newmap/search.py:464-544).  Results are those of nm_min_unique_one / nm_fixed_k_one.
// One lane sweeps one word of the need bitmap (64 positions).  A chain is a sequence of DEPENDENT reads of lines no other lane
// wants, so a turn of the state machine is shaped by two things: its latency -- ONE round of reads per turn, whatever the
// lane is doing (a step to the left: the LF entries of the interval's two ends; a walk's step: their rank blocks; the 32 LCP
// bytes around a row; the seed window; the two seed entries), so that the lanes of a wave stay in step -- and the number of
// load instructions, each of which costs the L1 one cycle per lane when every lane reads its own line (measured: with
// eight loads per turn the kernel ran at the L1's pace).  Hence the bases come from registers (the word's own planes, the
// walk's window), the superblock constants from `sc` (the caller's copy: LDS on the device), and the second block is read
// only where the interval leaves the first.
#define NM_SW_IDLE 0u
#define NM_SW_SEED 1u              /* read the window of the position's first bases */
#define NM_SW_SEED2 2u             /* read the seed entries of the window and of its reverse complement (slots in iv.k, iv.l) */
#define NM_SW_WALK 3u
#define NM_SW_LEFT 4u
#define NM_SW_LCP 5u               /* read the LCP bytes around the one row (iv.k) that is left of the chain's string */
#define NM_SW_VALID 1u             /* flags: iv holds the rows of S[64 word + qo .. 64 word + Fo), two or more */
#define NM_SW_EXACT 2u             /*        and S[64 word + qo .. 64 word + Fo + 1) occurs once */
#define NM_SW_EMIT 1u              /* step result: element `out_v` of position `out_p` is decided */
#define NM_SW_ERR 2u               /*              a k-mer of position `out_p` is absent from the index */
#define NM_SW_DONE 4u              /*              the word is finished */
#define NM_SW_NONE 0xFFFFFFFFu     /* no unique length as far as a walk may look */

struct nm_sweep_args {
    uint32_t kmin, kmax;           // list mode: the first and the longest listed length
    uint64_t seq_len;
    const uint32_t *list;          // nullptr: range mode
    uint32_t n_list;
    const uint64_t *sc;            // [n_super][4]: C[c] + occurrences of c before the superblock (ix.superC, or a nearer copy)
};

struct nm_sweep {
    uint64_t bits;                 // open positions of the word not yet taken
    uint64_t wlo, whi;             // planes of the word's own 64 bases
    nm_bi iv;                      // k: rows of the string, l: rows of its reverse complement
    nm_window w;                   // WALK: bases [p + kbase, p + kbase + 64) of the position being decided
    uint32_t word;                 // the word of the need bitmap being swept: positions 64 word ..
    uint32_t qo, Fo;               // the chain's string, offsets from the word's first position
    uint32_t flags, mode;
    uint32_t po, k, kbase;         // the position being decided; WALK: iv spells its first k bases
};

struct nm_q4 { uint64_t x[4]; };   // 32 bytes as loaded
NM_HD nm_q4 nm_load_q4(const void *p) {
    const uint64_t *q = (const uint64_t *)p;
    nm_q4 r;
    r.x[0] = q[0]; r.x[1] = q[1]; r.x[2] = q[2]; r.x[3] = q[3];
    return r;
}
// The two reads of a turn are issued back to back and waited for ONCE.  The device build spells them out (nm_engine.hip:
// written as plain loads, the compiler waits for the first block before it issues the second wherever the second is
// conditional); here, for the host mirror, they are plain loads.
#ifndef NM_Q4_ISSUE
typedef nm_q4 nm_q4_raw;
#define NM_Q4_ZERO(r) ((r).x[0] = (r).x[1] = (r).x[2] = (r).x[3] = 0)
#define NM_Q4_ISSUE(r, p) ((r) = nm_load_q4(p))
#define NM_Q2_ISSUE(r, p) ((r).x[0] = ((const uint64_t *)(p))[0], (r).x[1] = ((const uint64_t *)(p))[1], (r).x[2] = (r).x[3] = 0)
#define NM_Q4_WAIT2(a, b) ((void)0)
#define NM_Q4_VALUE(r) (r)
#endif
NM_HD nm_blk nm_blk_of(const nm_q4 &q) {
    nm_blk b;
    b.c0 = (uint32_t)q.x[0]; b.c1 = (uint32_t)(q.x[0] >> 32); b.c2 = (uint32_t)q.x[1]; b.c3 = (uint32_t)(q.x[1] >> 32);
    b.lo = q.x[2]; b.hi = q.x[3];
    return b;
}

NM_HD void nm_sweep_begin(nm_sweep &st, uint64_t word, uint64_t bits, uint64_t wlo, uint64_t whi) {
    st.word = (uint32_t)word; st.bits = bits; st.wlo = wlo; st.whi = whi;
    st.flags = 0; st.mode = NM_SW_IDLE;
    st.qo = st.Fo = st.po = st.k = st.kbase = 0;
    st.iv.k = st.iv.l = st.iv.s = 0;
    st.w.lo = st.w.hi = st.w.amb = 0;
}

// the element of a position whose least unique length is L (NM_SW_NONE: none within reach)
NM_HD uint32_t nm_sweep_element(const nm_enc_word *enc, const nm_sweep_args &a, uint64_t p, uint32_t L) {
    if (!a.list) return L > a.kmax ? 0u : (L > a.kmin ? L : a.kmin);       // (an open position has kmin unambiguous bases)
    // newmap/search.py:551-644: the first listed length whose (truncated, :590) k-mer occurs once; an ambiguous base inside
    // a listed k-mer drops the position for good (:593-596)
    const uint32_t U = nm_upper_one(enc, p, a.kmax);
    const uint64_t rem = a.seq_len - p;
    for (uint32_t i = 0; i < a.n_list; i++) {
        const uint32_t K = a.list[i];
        const uint32_t Lt = (uint64_t)K < rem ? K : (uint32_t)rem;
        if (Lt > U) return 0;
        if (L <= Lt) return K;
    }
    return 0;
}

template <bool BIG>
NM_HD uint32_t nm_sweep_step(const nm_view &ix, const nm_enc_word *enc, const nm_sweep_args &a, nm_sweep &st, uint64_t &out_p,
                             uint32_t &out_v, nm_tally &t) {
    if (st.mode == NM_SW_IDLE) {
        if (!st.bits) return NM_SW_DONE;
        const uint32_t o = 63u - (uint32_t)__builtin_clzll(st.bits);
        st.bits &= ~(1ULL << o);
        st.po = o;
        st.mode = ((st.flags & NM_SW_VALID) && st.qo == o + 1) ? NM_SW_LEFT : NM_SW_SEED;
    }
    const uint64_t p = (uint64_t)st.word * 64 + st.po;
    out_p = p;
    // ---- the turn's reads: addresses from the state alone, nothing used before all of them are on their way
    const uint32_t mode = st.mode;
    const bool walk = mode == NM_SW_WALK, ext = walk || mode == NM_SW_LEFT;
    const uint64_t row = walk ? st.iv.l : st.iv.k;         // the interval an extension reads (to the right: the reverse complement's)
    const uint64_t end = row + st.iv.s;
    const bool same = (row >> 6) == (end >> 6);
    uint32_t j = st.k - st.kbase;                          // WALK: the base the step takes, in the window
    const bool reload = walk && j >= 64;                   // (a walk longer than its window: the next 64 bases)
    const bool window = mode == NM_SW_SEED || reload;
    const uint64_t wat = reload ? p + st.k : p;            // first base of the window to read
    const uint64_t lcp0 = st.iv.k > 12 ? (st.iv.k - 12) & ~3ULL : 0ULL;      // LCP: the 32 bytes from row lcp0 on (12 .. 15 rows before the row, 16 .. 19 after)
    // a step to the LEFT needs the rows of the string alone (its reverse complement's only matter to a walk, and a walk starts
    // from the seed table): where the LF blocks were built it is a plain LF step on ONE 16-byte entry per end -- half the
    // requests of a rank block, a third of the arithmetic
    const uint32_t lcode = (uint32_t)((st.wlo >> st.po) & 1ULL) | ((uint32_t)((st.whi >> st.po) & 1ULL) << 1);
    const bool left_lf = mode == NM_SW_LEFT && ix.lfb != nullptr;
    const void *pa, *pb;
    if (window) { pa = enc + (wat >> 6); pb = enc + (wat >> 6) + 1; }
    else if (left_lf) { pa = ix.lfb + ((row >> 6) * 4 + lcode); pb = ix.lfb + ((end >> 6) * 4 + lcode); }
    else if (ext) { pa = ix.rank + (row >> 6); pb = ix.rank + (end >> 6); }
    else if (mode == NM_SW_LCP) { pa = ix.lcp + lcp0; pb = pa; }
    else { pa = ix.seed + (st.iv.l & ~3ULL); pb = ix.seed + (st.iv.k & ~3ULL); }      // SEED2: the aligned four entries that hold the slot
    const bool one = (ext && !reload && same) || mode == NM_SW_LCP;          // (the second rank block only where the interval leaves the first)
    nm_q4_raw ra, rb;
    NM_Q4_ZERO(rb);
    if (left_lf) {
        NM_Q2_ISSUE(ra, pa);
        if (!one) NM_Q2_ISSUE(rb, pb);
    } else {
        NM_Q4_ISSUE(ra, pa);
        if (!one) NM_Q4_ISSUE(rb, pb);
    }
    NM_Q4_WAIT2(ra, rb);                                   // ONE wait per turn, after every read of the turn is on its way
    const nm_q4 A = NM_Q4_VALUE(ra), B = NM_Q4_VALUE(rb);
    // ---- and what they mean
    if (window) {
        const uint32_t sh = (uint32_t)(wat & 63);
        st.w.lo = A.x[0]; st.w.hi = A.x[1]; st.w.amb = A.x[2];
        if (sh) { st.w.lo = (st.w.lo >> sh) | (B.x[0] << (64 - sh)); st.w.hi = (st.w.hi >> sh) | (B.x[1] << (64 - sh)); st.w.amb = (st.w.amb >> sh) | (B.x[2] << (64 - sh)); }
        if (reload) { st.kbase = st.k; return 0; }
        const uint32_t sl = ix.seed_len;
        st.flags = 0;
        st.kbase = 0;
        if (ix.seed && sl && a.kmin >= sl && (st.w.amb & ((1ULL << sl) - 1ULL)) == 0) {
            st.iv.l = nm_seed_slot(st.w, sl);              // an entry holds the rows of the reverse complement of its slot's string
            st.iv.k = nm_slot_revcomp(st.iv.l, sl);
            st.mode = NM_SW_SEED2;
        } else {                                           // no seed table for this search (or, list mode, fewer unambiguous bases): from the first base
            st.iv.k = 0; st.iv.l = 0; st.iv.s = ix.n; st.k = 0;
            st.mode = NM_SW_WALK;
        }
        return 0;
    }
    if (mode == NM_SW_LCP) {
        // The chain's string, one base longer to the left -- S[p .. F), len = F - p bases -- has ONE row among the rows the chain
        // holds: r = iv.k.  The longest prefix the suffix of r shares with another suffix is the larger of its two LCP bytes, l.
        //   l < len: the string really occurs once; one base more than l and a prefix of it occurs once -- the least unique length
        //            of the position, no walk -- and the rows around r whose bytes reach l are the rows of that prefix.
        //   l >= len: the string occurs twice after all (the chain held only SOME of its rows, see below): the end has not
        //            moved, the position is decided as by a step that found two rows; the rows whose bytes reach len are its rows.
        // The chain goes on from those rows -- from as many of them as the 32 bytes read here show: a subset of a string's rows
        // can only under-count, and an under-count ends up here again, where the bytes put it right.  (The reverse complement's
        // rows are not known any more: only walks, which start from the seed table, need them.)
        const uint64_t r = st.iv.k;
        const uint32_t len = st.Fo - st.po, was = st.k;    // (the flags of the chain the step came from, kept in k)
        auto byte_at = [&](uint64_t row_) -> uint32_t {
            const uint32_t o = (uint32_t)(row_ - lcp0);
            const uint64_t wd = (o >> 3) == 0 ? A.x[0] : ((o >> 3) == 1 ? A.x[1] : ((o >> 3) == 2 ? A.x[2] : A.x[3]));
            return (uint32_t)(wd >> (8 * (o & 7u))) & 0xFFu;
        };
        const uint32_t b0 = byte_at(r), b1 = byte_at(r + 1);
        const uint32_t l = b0 > b1 ? b0 : b1;
        t.blocks++;
        if (l >= NM_LCP_CAP && (len > NM_LCP_CAP || a.kmax > NM_LCP_CAP)) {      // a capped byte that decides nothing here: the position walks for itself
            st.flags = 0;
            st.mode = NM_SW_SEED;
            return 0;
        }
        const bool moved = l < len;                        // (l capped: len <= 255 here, so l >= len; kmax <= 255: every length in question is above kmax)
        const uint32_t reach = moved ? l : len;
        uint64_t lo = r, hi = r + 1;
        while (lo > lcp0 && byte_at(lo) >= reach) lo--;
        while (hi < lcp0 + 31 && byte_at(hi) >= reach) hi++;
        st.iv.k = lo; st.iv.l = 0; st.iv.s = hi - lo;
        st.qo = st.po;
        if (moved) { st.Fo = st.po + l; st.flags = NM_SW_VALID | NM_SW_EXACT; }
        else st.flags = was;
        st.mode = NM_SW_IDLE;
        out_v = nm_sweep_element(enc, a, p, moved ? l + 1 : ((was & NM_SW_EXACT) ? len + 1 : NM_SW_NONE));
        if (st.iv.s < 2) st.flags = 0;                     // (cannot happen: one neighbour's byte reaches `reach`; then the next position walks)
        return NM_SW_EMIT;
    }
    if (mode == NM_SW_SEED2) {
        uint64_t rlo, rhi, flo, fhi;
        t.seeds += 2;
        const uint32_t ir = (uint32_t)(st.iv.l & 3), jf = (uint32_t)(st.iv.k & 3);     // (selects: an indexed read would put the loads into scratch)
        const uint64_t er = ir == 0 ? A.x[0] : (ir == 1 ? A.x[1] : (ir == 2 ? A.x[2] : A.x[3]));
        const uint64_t ef = jf == 0 ? B.x[0] : (jf == 1 ? B.x[1] : (jf == 2 ? B.x[2] : B.x[3]));
        if (nm_seed_decode(er, rlo, rhi) && nm_seed_decode(ef, flo, fhi) && rhi - rlo >= 2 && fhi - flo == rhi - rlo) {
            st.iv.k = flo; st.iv.l = rlo; st.iv.s = rhi - rlo;
            st.k = ix.seed_len;
        } else {                                           // saturated, or fewer than two rows: walk from the first base
            st.iv.k = 0; st.iv.l = 0; st.iv.s = ix.n; st.k = 0;
        }
        st.mode = NM_SW_WALK;
        return 0;
    }
    // ONE extension: to the right (WALK: appending a base = prepending its complement to the reverse complement) or to the left
    const uint32_t code = walk ? 3u - nm_window_code(st.w, j) : lcode;
    if (walk && (st.k >= a.kmax || ((st.w.amb >> j) & 1ULL))) {      // nothing up to kmax / up to U_p occurs once
        st.qo = st.po; st.Fo = st.po + st.k;
        st.flags = NM_SW_VALID;
        st.mode = NM_SW_IDLE;
        out_v = nm_sweep_element(enc, a, p, NM_SW_NONE);
        return NM_SW_EMIT;
    }
    nm_bi x;
    x.k = row; x.l = walk ? st.iv.k : st.iv.l; x.s = st.iv.s;
    t.steps++;
    t.blocks += same ? 1u : 2u;
    if (left_lf) {
        const nm_q4 &Bh = one ? A : B;
        const uint64_t lo2 = A.x[0] + nm_popc64(A.x[1] & ((1ULL << (row & 63)) - 1ULL));
        const uint64_t hi2 = Bh.x[0] + nm_popc64(Bh.x[1] & ((1ULL << (end & 63)) - 1ULL));
        x.k = lo2; x.l = 0; x.s = hi2 - lo2;
    } else {
        const uint64_t *cp = BIG ? a.sc + (row >> NM_SUPER_SHIFT) * 4 : ix.C;
        const uint64_t cs[4] = {cp[0], cp[1], cp[2], cp[3]};
        nm_bi_extend_blk<BIG>(ix, x, code, nm_blk_of(A), nm_blk_of(one ? A : B), cs);
    }
    if (x.s >= 2) {
        st.iv.k = walk ? x.l : x.k; st.iv.l = walk ? x.k : x.l; st.iv.s = x.s;
        if (walk) { st.k++; return 0; }
        st.qo = st.po;                                     // LEFT: decided by the one step
        st.mode = NM_SW_IDLE;
        out_v = nm_sweep_element(enc, a, p, (st.flags & NM_SW_EXACT) ? st.Fo + 1 - st.po : NM_SW_NONE);
        return NM_SW_EMIT;
    }
    if (!walk) {                                           // LEFT: the end moved
        if (ix.lcp && x.s == 1) { st.iv.k = x.k; st.k = st.flags; st.flags = 0; st.mode = NM_SW_LCP; return 0; }      // one row left: its LCP bytes say where the new end is
        st.flags = 0;
        st.mode = NM_SW_SEED;                              // (no LCP bytes in this index, or a string that is absent): the position walks for itself
        return 0;
    }
    st.mode = NM_SW_IDLE;
    if (x.s == 0) {                                        // search.py:699-722
        st.flags = 0;
        out_v = 0;
        return NM_SW_EMIT | NM_SW_ERR;
    }
    st.qo = st.po; st.Fo = st.po + st.k;                   // S[p .. p + k) still occurs twice, one base more and it occurs once
    st.flags = NM_SW_VALID | NM_SW_EXACT;
    out_v = nm_sweep_element(enc, a, p, st.k + 1);
    return NM_SW_EMIT;
}

// Which strides get a repeat probe when the probes run AFTER the sites: a stretch that occurs twice over more than a
// stride leaves (nearly) all its positions unsettled, so a probe is worth its walk only where the stride itself or
// the stride before it (whose positions read this stride's word, nm_probe_kstar) is mostly unsettled.  A stride
// without a probe has the word 0: nothing decided, its positions search for themselves.
#define NM_PROBE_GATE_BITS 32u
NM_HD bool nm_probe_gate(const uint64_t *need, uint64_t j, uint64_t n_words) {
    if (j < n_words && nm_popc64(need[j]) >= NM_PROBE_GATE_BITS) return true;
    return j > 0 && j - 1 < n_words && nm_popc64(need[j - 1]) >= NM_PROBE_GATE_BITS;
}

#endif

// nm_hash.h -- record fingerprints (index format 2): is the sequence being searched one of the indexed records?
//
// The reference stops a search when a probed k-mer is absent from the index ("possibly a mismatch between the sequence
// and the index", newmap/search.py:699-722).  Whether that can happen at all is decided per record: a record whose bases
// are, position by position, those of an indexed record contains no absent k-mer, so its search needs no check; any
// other record is searched with every probe of the reference's schedule verified (nm_core.h: nm_guard_*).
//
// The fingerprint is taken over the ENCODED record -- the index's text model: case folded, one ambiguity class
// (nm_format.h) -- 64 bases at a time: word W of a record holds the bit-planes lo / hi (2-bit base codes, A = 0 .. T = 3)
// and amb (1 = not ACGT; lo = hi = 0 there) of bases [64 W, 64 W + 64), zero past the record's end.
//
//      t(W) = mix(mix(mix(lo + K) ^ hi) ^ amb)          mix = a 64-bit bijection (splitmix64 finaliser)
//      H    = sum over W of t(W) * R^W   (mod 2^64),    R = NM_HASH_R (odd)
//
// so the H of a record is the sum of R^(s / 64) * H(segment) over its segments [s, s + n) when every s is a multiple of
// 64: a record is fingerprinted segment by segment, on the device, by the kernel that encodes it (k_sites; k_segment_hash
// on the other paths), and the segments' sums are joined on the host.  A record matches an indexed one when length and H
// agree: different (lo, hi, amb) triples of a word collide with probability 2^-64, and a record that differs in one word
// only always differs in H (R^W is odd).  The index file lists (length, H) of its records.
#ifndef NM_HASH_H
#define NM_HASH_H

#include <stdint.h>

#define NM_HASH_R 0x9E3779B97F4A7C15ULL
#define NM_HASH_K 0xD6E8FEB86659FD93ULL

#ifndef NM_HASH_FN
#define NM_HASH_FN static inline
#endif

NM_HASH_FN uint64_t nm_hash_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// t(W): the planes of one 64-base word (bases past the end of the record / outside the positions taken: zero bits)
NM_HASH_FN uint64_t nm_hash_word(uint64_t lo, uint64_t hi, uint64_t amb) {
    return nm_hash_mix(nm_hash_mix(nm_hash_mix(lo + NM_HASH_K) ^ hi) ^ amb);
}

NM_HASH_FN uint64_t nm_hash_pow(uint64_t e) {              // R^e mod 2^64
    uint64_t r = 1, b = NM_HASH_R;
    while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
    return r;
}

// H(a . b) from H(a), the number of 64-base words of a (len(a) a multiple of 64) and H(b)
NM_HASH_FN uint64_t nm_hash_join(uint64_t ha, uint64_t words_a, uint64_t hb) { return ha + nm_hash_pow(words_a) * hb; }

// power table for the kernels: tab[256 j + t] = R^(t * 256^j), j = 0 .. 3  (word indexes below 2^32)
#define NM_HASH_TAB_WORDS 1024
NM_HASH_FN void nm_hash_fill_tables(uint64_t *tab) {
    uint64_t base = NM_HASH_R;
    for (int j = 0; j < 4; j++) {
        tab[256 * j] = 1;
        for (int t = 1; t < 256; t++) tab[256 * j + t] = tab[256 * j + t - 1] * base;
        base = tab[256 * j + 255] * base;                  // R^(256^(j+1))
    }
}
NM_HASH_FN uint64_t nm_hash_word_power(const uint64_t *tab, uint64_t w) {     // R^w, w < 2^32
    return tab[w & 255u] * tab[256 + ((w >> 8) & 255u)] * tab[512 + ((w >> 16) & 255u)] * tab[768 + ((w >> 24) & 255u)];
}

#endif

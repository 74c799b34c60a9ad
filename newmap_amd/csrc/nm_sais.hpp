// nm_sais.hpp -- suffix array construction by induced sorting (SA-IS; Nong, Zhang & Chan,
// "Two Efficient Algorithms for Linear Time Suffix Array Construction", IEEE TC 2011), written
// from the paper for the newmap_amd host index builder.  Linear time, recursion runs inside the
// output array, index type templated so that texts beyond 2^31 symbols use 64-bit entries.
//
// Requirements: n >= 1, s[n-1] == 0 and 0 occurs nowhere else (unique smallest terminator),
// every symbol < K.  I must be a signed integer type.
#ifndef NM_SAIS_HPP
#define NM_SAIS_HPP

#include <cstdint>
#include <vector>

namespace nm {

class TypeBits {                      // one bit per text position: 1 = S-type
public:
    explicit TypeBits(uint64_t n) : w_((n + 63) / 64, 0) {}
    inline bool get(uint64_t i) const { return (w_[i >> 6] >> (i & 63)) & 1; }
    inline void set(uint64_t i, bool v) {
        uint64_t m = 1ULL << (i & 63);
        if (v) w_[i >> 6] |= m; else w_[i >> 6] &= ~m;
    }
private:
    std::vector<uint64_t> w_;
};

template <class Ch, class I>
static void sais_buckets(const Ch *s, std::vector<I> &bkt, I n, I K, bool ends) {
    for (I c = 0; c < K; c++) bkt[c] = 0;
    for (I i = 0; i < n; i++) bkt[(I)s[i]]++;
    I sum = 0;
    for (I c = 0; c < K; c++) {
        sum += bkt[c];
        bkt[c] = ends ? sum : sum - bkt[c];
    }
}

template <class Ch, class I>
static void sais_induce(const Ch *s, I *SA, I n, I K, const TypeBits &t, std::vector<I> &bkt) {
    // L-type suffixes, left to right
    sais_buckets(s, bkt, n, K, false);
    for (I i = 0; i < n; i++) {
        I j = SA[i];
        if (j > 0 && !t.get((uint64_t)(j - 1))) SA[bkt[(I)s[j - 1]]++] = j - 1;
    }
    // S-type suffixes, right to left
    sais_buckets(s, bkt, n, K, true);
    for (I i = n - 1; i >= 0; i--) {
        I j = SA[i];
        if (j > 0 && t.get((uint64_t)(j - 1))) SA[--bkt[(I)s[j - 1]]] = j - 1;
    }
}

template <class Ch, class I>
void sais(const Ch *s, I *SA, I n, I K) {
    if (n == 1) { SA[0] = 0; return; }
    TypeBits t((uint64_t)n);
    t.set((uint64_t)(n - 1), true);
    for (I i = n - 2; i >= 0; i--)
        t.set((uint64_t)i, s[i] < s[i + 1] || (s[i] == s[i + 1] && t.get((uint64_t)(i + 1))));
    auto is_lms = [&](I i) { return i > 0 && t.get((uint64_t)i) && !t.get((uint64_t)(i - 1)); };

    std::vector<I> bkt((size_t)K);

    // stage 1: sort the LMS substrings
    sais_buckets(s, bkt, n, K, true);
    for (I i = 0; i < n; i++) SA[i] = -1;
    for (I i = 1; i < n; i++)
        if (is_lms(i)) SA[--bkt[(I)s[i]]] = i;
    sais_induce(s, SA, n, K, t, bkt);

    I n1 = 0;
    for (I i = 0; i < n; i++)
        if (is_lms(SA[i])) SA[n1++] = SA[i];
    for (I i = n1; i < n; i++) SA[i] = -1;
    I name = 0, prev = -1;
    for (I i = 0; i < n1; i++) {
        I pos = SA[i];
        bool diff = false;
        for (I d = 0; d < n; d++) {
            if (prev == -1 || s[pos + d] != s[prev + d] ||
                t.get((uint64_t)(pos + d)) != t.get((uint64_t)(prev + d))) { diff = true; break; }
            if (d > 0 && (is_lms(pos + d) || is_lms(prev + d))) break;
        }
        if (diff) { name++; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    for (I i = n - 1, j = n - 1; i >= n1; i--)
        if (SA[i] >= 0) SA[j--] = SA[i];

    // stage 2: order of the LMS suffixes = suffix array of the reduced string
    I *SA1 = SA, *s1 = SA + n - n1;
    if (name < n1) sais<I, I>(s1, SA1, n1, name);
    else for (I i = 0; i < n1; i++) SA1[s1[i]] = i;

    // stage 3: induce the full suffix array from the sorted LMS suffixes
    sais_buckets(s, bkt, n, K, true);
    for (I i = 1, j = 0; i < n; i++)
        if (is_lms(i)) s1[j++] = i;
    for (I i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
    for (I i = n1; i < n; i++) SA[i] = -1;
    for (I i = n1 - 1; i >= 0; i--) {
        I j = SA[i];
        SA[i] = -1;
        SA[--bkt[(I)s[j]]] = j;
    }
    sais_induce(s, SA, n, K, t, bkt);
}

}  // namespace nm
#endif

// nm_pdsa.hpp -- parallel suffix array construction by prefix doubling (Manber & Myers 1993;
// Larsson & Sadakane, "Faster suffix sorting", 2007), OpenMP, for the newmap_amd host index
// builder.  SA-IS (nm_sais.hpp) is linear but serial: ~26 s for the 100 Mbp benchmark genome and
// far too slow for a both-strand human genome (6.2 G symbols).  This sorter uses every host core:
//
//   1. every suffix gets a 63-bit key = its first 21 symbols (3 bits each); one parallel bucket
//      scatter on the leading bits plus independent per-bucket sorts order all suffixes by 21
//      symbols -- for DNA that already resolves everything outside repeats;
//   2. suffixes that still share a key form groups; rank[i] = first index of i's group;
//   3. while groups remain: sort each group by rank[i + h] (two-phase: all groups are sorted against
//      the ranks of the previous round, then the ranks are updated), split it into sub-groups,
//      h doubles.  Deterministic, independent of the thread count, O(n log maxLCP).
//
// Requirements as for nm::sais: s[n-1] == 0 is the unique smallest symbol, all symbols < 8.
// The result is the true lexicographic suffix array, hence identical to nm::sais's.
#ifndef NM_PDSA_HPP
#define NM_PDSA_HPP

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace nm {

static inline double pd_now() {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}
static inline bool pd_verbose() { const char *v = getenv("NEWMAP_AMD_VERBOSE"); return v && *v && *v != '0'; }

static inline int pd_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// uninitialised array: large buffers are first touched by the parallel loops that fill them
template <class T>
class PdBuf {
public:
    explicit PdBuf(uint64_t n) : p_((T *)malloc((size_t)(n ? n : 1) * sizeof(T))), bytes_((size_t)(n ? n : 1) * sizeof(T)) { if (!p_) throw std::bad_alloc(); }
    ~PdBuf() { release(); }
    PdBuf(const PdBuf &) = delete;
    PdBuf &operator=(const PdBuf &) = delete;
    T *data() { return p_; }
    T &operator[](uint64_t i) { return p_[i]; }
    // unmapping 100 GB of touched pages takes ~10 s: big buffers are handed to a detached thread so that the
    // sorter goes on while the kernel takes the pages back
    void release() {
        if (p_ && bytes_ >= ((size_t)1 << 30)) {
            T *p = p_;
            try { std::thread([p] { free(p); }).detach(); } catch (...) { free(p); }
        } else {
            free(p_);
        }
        p_ = nullptr;
    }
    // first touch by all threads, each on its own contiguous slice: page faults are then taken in
    // parallel and sequentially instead of inside a random scatter
    void touch(uint64_t n) {
        const int nt = pd_threads();
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            const uint64_t a = n * (uint64_t)t / nt, b = n * (uint64_t)(t + 1) / nt;
            if (b > a) memset((void *)(p_ + a), 0, (size_t)(b - a) * sizeof(T));
        }
    }
private:
    T *p_;
    size_t bytes_;
};

// (key, position) pair of the initial sort
template <class I>
struct PdKI {
    uint64_t k;
    I i;
    bool operator<(const PdKI &o) const { return k != o.k ? k < o.k : i < o.i; }
};

// Parallel sort of `a` (n pairs) by (key, position): one scatter into 2^PD_TOP_BITS buckets on the
// leading key bits, then an independent comparison sort inside every bucket (cache-sized for DNA).
// `b` is scratch of the same size; the sorted data ends in `a`.
#define PD_TOP_BITS 18
template <class I>
static void pd_bucket_sort(PdKI<I> *a, PdKI<I> *b, uint64_t n, unsigned key_bits) {
    const int nt = pd_threads();
    const unsigned shift = key_bits > PD_TOP_BITS ? key_bits - PD_TOP_BITS : 0;
    const size_t nb = (size_t)1 << PD_TOP_BITS;
    std::vector<uint64_t> hist((size_t)nt * nb, 0);
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const uint64_t lo = n * (uint64_t)t / nt, hi = n * (uint64_t)(t + 1) / nt;
        uint64_t *h = hist.data() + (size_t)t * nb;
        for (uint64_t i = lo; i < hi; i++) h[a[i].k >> shift]++;
    }
    std::vector<uint64_t> start(nb + 1, 0);
    uint64_t sum = 0;
    for (size_t d = 0; d < nb; d++) {
        start[d] = sum;
        for (int t = 0; t < nt; t++) {
            const uint64_t c = hist[(size_t)t * nb + d];
            hist[(size_t)t * nb + d] = sum;
            sum += c;
        }
    }
    start[nb] = sum;
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const uint64_t lo = n * (uint64_t)t / nt, hi = n * (uint64_t)(t + 1) / nt;
        uint64_t *h = hist.data() + (size_t)t * nb;
        for (uint64_t i = lo; i < hi; i++) b[h[a[i].k >> shift]++] = a[i];
    }
#pragma omp parallel for schedule(dynamic, 8) num_threads(nt)
    for (int64_t d = 0; d < (int64_t)nb; d++) {
        const uint64_t s0 = start[d], s1 = start[d + 1];
        if (s1 - s0 > 1) std::sort(b + s0, b + s1);
    }
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t i = 0; i < (int64_t)n; i++) a[i] = b[i];
}

template <class I>
struct PdGroup { I start, len; };

// SA must hold n entries of an unsigned or signed integer type wide enough for n.
template <class I>
void pd_suffix_array(const uint8_t *s, uint64_t n, I *SA) {
    if (n == 0) return;
    if (n == 1) { SA[0] = 0; return; }
    const int nt = pd_threads();
    const unsigned H0 = 21;                                  // symbols in the initial key
    const bool verbose = pd_verbose();
    double t0 = pd_now();
    PdBuf<uint64_t> K(n);
    PdBuf<I> V2(n);
    {
        PdBuf<PdKI<I>> A(n), B(n);
        // ---- 1. keys: first symbol in the most significant position, zero (= terminator) padded
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            const uint64_t a = n * (uint64_t)t / nt, b = n * (uint64_t)(t + 1) / nt;
            if (a < b) {
                uint64_t key = 0;                            // key of position b (or 0 past the end)
                for (unsigned j = 0; j < H0; j++) {
                    const uint64_t p = b + j;
                    key |= (uint64_t)(p < n ? s[p] : 0) << (3 * (H0 - 1 - j));
                }
                for (uint64_t i = b; i-- > a;) {
                    key = (key >> 3) | ((uint64_t)s[i] << (3 * (H0 - 1)));
                    A[i].k = key;
                    A[i].i = (I)i;
                }
            }
        }
        if (verbose) { fprintf(stderr, "[pdsa] keys %.2fs\n", pd_now() - t0); t0 = pd_now(); }
        pd_bucket_sort<I>(A.data(), B.data(), n, 3 * H0);
        if (verbose) { fprintf(stderr, "[pdsa] initial sort %.2fs\n", pd_now() - t0); t0 = pd_now(); }
#pragma omp parallel for schedule(static) num_threads(nt)
        for (int64_t j = 0; j < (int64_t)n; j++) { K[j] = A[j].k; SA[j] = A[j].i; }
        if (verbose) { fprintf(stderr, "[pdsa]   keys and positions apart %.2fs\n", pd_now() - t0); }
    }
    if (verbose) { fprintf(stderr, "[pdsa]   sort buffers released %.2fs\n", pd_now() - t0); }

    // ---- 2. group heads and ranks
    PdBuf<I> R(n);                                           // R[i] = index of the head of i's group
    R.touch(n);
    if (verbose) { fprintf(stderr, "[pdsa]   rank array touched %.2fs\n", pd_now() - t0); }
    PdBuf<I> &H = V2;                                        // H[j] = head index of sorted position j
    std::vector<std::vector<PdGroup<I>>> tg((size_t)nt);
    {
        std::vector<uint64_t> last_head((size_t)nt, 0);
        std::vector<uint8_t> has_head((size_t)nt, 0);
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            const uint64_t a = n * (uint64_t)t / nt, b = n * (uint64_t)(t + 1) / nt;
            uint64_t head = 0;
            bool seen = false;
            for (uint64_t j = a; j < b; j++) {
                if (j == 0 || K[j] != K[j - 1]) { head = j; seen = true; }
                H[j] = (I)head;                              // provisional inside a chunk-crossing group
            }
            last_head[t] = head;
            has_head[t] = seen;
        }
        if (verbose) { fprintf(stderr, "[pdsa]   heads %.2fs\n", pd_now() - t0); }
        // groups that cross chunk starts: carry the head from the left
        std::vector<uint64_t> carry((size_t)nt, 0);
        for (int t = 1; t < nt; t++) carry[t] = has_head[t - 1] ? last_head[t - 1] : carry[t - 1];
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            const uint64_t a = n * (uint64_t)t / nt, b = n * (uint64_t)(t + 1) / nt;
            for (uint64_t j = a; j < b; j++) {
                if (j == 0 || K[j] != K[j - 1]) break;       // first real head of this chunk
                H[j] = (I)carry[t];
            }
        }
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            const uint64_t a = n * (uint64_t)t / nt, b = n * (uint64_t)(t + 1) / nt;
            for (uint64_t j = a; j < b; j++) R[(uint64_t)SA[j]] = H[j];
            // unresolved groups whose head lies in this chunk
            for (uint64_t j = a; j < b;) {
                if ((uint64_t)H[j] != j) { j++; continue; }  // not a head (belongs to a group from the left)
                uint64_t e = j + 1;
                while (e < n && (uint64_t)H[e] == j) e++;
                if (e - j > 1) tg[t].push_back(PdGroup<I>{(I)j, (I)(e - j)});
                j = e;
            }
        }
    }
    K.release();
    std::vector<PdGroup<I>> groups;
    for (auto &v : tg) { groups.insert(groups.end(), v.begin(), v.end()); v.clear(); }

    if (verbose) { fprintf(stderr, "[pdsa] ranks + groups %.2fs, %zu unresolved groups\n", pd_now() - t0, groups.size()); t0 = pd_now(); }
    // ---- 3. doubling rounds
    typedef std::pair<I, I> KV;                              // (rank of i+h, i)
    for (uint64_t h = H0; !groups.empty(); h *= 2) {
        const int64_t ng = (int64_t)groups.size();
        // phase A: sort every group by the rank h symbols further on; H[j] <- new head of position j
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            std::vector<KV> buf;
#pragma omp for schedule(dynamic, 16)
            for (int64_t g = 0; g < ng; g++) {
                const uint64_t st = (uint64_t)groups[g].start, len = (uint64_t)groups[g].len;
                buf.resize(len);
                for (uint64_t j = 0; j < len; j++) {
                    const uint64_t i = (uint64_t)SA[st + j];
                    buf[j] = KV(R[i + h], (I)i);             // i + h < n: a group member has not reached '#'
                }
                std::sort(buf.begin(), buf.end());
                uint64_t head = 0;
                for (uint64_t j = 0; j < len; j++) {
                    if (j && buf[j].first != buf[j - 1].first) {
                        if (j - head > 1) tg[t].push_back(PdGroup<I>{(I)(st + head), (I)(j - head)});
                        head = j;
                    }
                    SA[st + j] = buf[j].second;
                    H[st + j] = (I)(st + head);
                }
                if (len - head > 1) tg[t].push_back(PdGroup<I>{(I)(st + head), (I)(len - head)});
            }
        }
        // phase B: publish the new ranks
#pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
        for (int64_t g = 0; g < ng; g++) {
            const uint64_t st = (uint64_t)groups[g].start, len = (uint64_t)groups[g].len;
            for (uint64_t j = 0; j < len; j++) R[(uint64_t)SA[st + j]] = H[st + j];
        }
        groups.clear();
        for (auto &v : tg) { groups.insert(groups.end(), v.begin(), v.end()); v.clear(); }
        // keep the work list in index order so that the result never depends on scheduling
        std::sort(groups.begin(), groups.end(),
                  [](const PdGroup<I> &x, const PdGroup<I> &y) { return x.start < y.start; });
        if (verbose) { fprintf(stderr, "[pdsa] round h=%llu: %.2fs, %zu groups left\n", (unsigned long long)h, pd_now() - t0, groups.size()); t0 = pd_now(); }
    }
}

}  // namespace nm
#endif

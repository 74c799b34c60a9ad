// nm_fasta_scan.hpp -- the host-side FASTA scan of the native driver's parallel front-end (csrc/nm_driver.hip):
// a mapped file -> records (ids, data byte ranges), pieces of every record cut at line starts, the number of bases
// in front of each piece, and the strip of a range of pieces into a buffer.  Plain C++17 + threads: the test-only
// host simulator (tests/hostsim) includes it too, so that the CPU suite checks these rules
// (newmap/fasta.py:20-190: header lines start with '>' or ';', id = first token minus its first byte, every line
// loses its trailing whitespace, data in front of any header has the id "", a record without data yields nothing)
// against the Python reader and the reference's fixtures without a GPU.
#ifndef NM_FASTA_SCAN_HPP
#define NM_FASTA_SCAN_HPP

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace nm_fasta {

inline bool is_space(unsigned char c) {      // what bytes.rstrip() removes (newmap/fasta.py:47)
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

// run fn(i) for i in [0, n) on up to `threads` threads (dynamic distribution)
inline void parallel_for(size_t n, unsigned threads, const std::function<void(size_t)> &fn) {
    if (n == 0) return;
    if (threads <= 1 || n == 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < n;) fn(i); };
    std::vector<std::thread> pool;
    const unsigned t = threads < n ? threads : (unsigned)n;
    for (unsigned k = 1; k < t; k++) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
}

inline unsigned host_threads() {
    if (const char *e = getenv("NEWMAP_AMD_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned)v; }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw == 0 ? 4 : (hw > 32 ? 32 : hw);
}

struct Record {
    std::string id;
    const unsigned char *data = nullptr, *end = nullptr;   // its data lines in the mapped file
    std::vector<const unsigned char *> chunk;             // line starts that cut the data into pieces of ~chunk_bytes
    std::vector<uint64_t> before;                          // bases (stripped bytes) in front of each piece; back() = all of them
    uint64_t n_bases = 0;
    int file = -1;                                         // output file (run of adjacent records with one id), -1 = none
    uint64_t file_offset = 0;                              // first element of this record inside that file
    uint64_t global = 0;                                   // first position of this record in the position space of the job
};

// bases of the lines in [p, e): every line without its trailing whitespace (bytes.rstrip(), newmap/fasta.py:47);
// dst != nullptr: copy them there.  p is a line start.
inline uint64_t strip_lines(const unsigned char *p, const unsigned char *e, uint8_t *dst) {
    uint64_t n = 0;
    while (p < e) {
        const unsigned char *nl = (const unsigned char *)memchr(p, '\n', (size_t)(e - p));
        const unsigned char *le = nl ? nl : e;
        const unsigned char *q = le;
        while (q > p && is_space(q[-1])) q--;
        const size_t len = (size_t)(q - p);
        if (dst && len) memcpy(dst + n, p, len);
        n += len;
        p = nl ? nl + 1 : e;
    }
    return n;
}

// the records of a mapped FASTA file, their pieces and base counts (threaded)
static const size_t kPieceBytes = 512u << 10;             // a record's data lines are cut into pieces of about this many file bytes
inline std::vector<Record> scan(const unsigned char *base, size_t size, unsigned threads, size_t chunk_bytes = kPieceBytes) {
    // ---- header lines: '>' or ';' at a line start (newmap/fasta.py:59), found by a threaded scan
    std::vector<size_t> heads;
    {
        const size_t slab = chunk_bytes * 8;
        const size_t n_slabs = (size + slab - 1) / slab;
        std::vector<std::vector<size_t>> found(n_slabs);
        parallel_for(n_slabs, threads, [&](size_t k) {
            const size_t lo = k * slab, hi = lo + slab < size ? lo + slab : size;
            for (const char c : {'>', ';'}) {
                const unsigned char *p = base + lo;
                while (p < base + hi) {
                    p = (const unsigned char *)memchr(p, c, (size_t)(base + hi - p));
                    if (!p) break;
                    const size_t i = (size_t)(p - base);
                    if (i == 0 || base[i - 1] == '\n') found[k].push_back(i);
                    p++;
                }
            }
        });
        for (auto &v : found) heads.insert(heads.end(), v.begin(), v.end());
        std::sort(heads.begin(), heads.end());
    }
    // ---- records (ids: first whitespace-delimited token minus its first byte, newmap/fasta.py:75; data in front of any
    // header has the id "")
    std::vector<Record> recs;
    auto add_record = [&](const std::string &id, size_t lo, size_t hi) {
        Record r;
        r.id = id;
        r.data = base + lo;
        r.end = base + hi;
        recs.push_back(std::move(r));
    };
    if (heads.empty() || heads[0] > 0) add_record("", 0, heads.empty() ? size : heads[0]);
    for (size_t h = 0; h < heads.size(); h++) {
        const unsigned char *p = base + heads[h];
        const unsigned char *nl = (const unsigned char *)memchr(p, '\n', size - heads[h]);
        size_t len = nl ? (size_t)(nl - p) : size - heads[h];
        while (len && is_space(p[len - 1])) len--;
        size_t e = 0;
        while (e < len && !is_space(p[e])) e++;
        const size_t data_lo = nl ? (size_t)(nl - base) + 1 : size;
        add_record(std::string((const char *)p + 1, e ? e - 1 : 0), data_lo, h + 1 < heads.size() ? heads[h + 1] : size);
    }
    // ---- pieces of every record and the bases in front of each (threaded count)
    struct Piece { size_t rec, idx; };
    std::vector<Piece> pieces;
    for (size_t r = 0; r < recs.size(); r++) {
        Record &rec = recs[r];
        const unsigned char *p = rec.data;
        while (p < rec.end) {
            rec.chunk.push_back(p);
            const unsigned char *q = p + chunk_bytes < rec.end ? p + chunk_bytes : rec.end;
            if (q < rec.end) {                                  // cut at the next line start
                const unsigned char *nl = (const unsigned char *)memchr(q, '\n', (size_t)(rec.end - q));
                q = nl ? nl + 1 : rec.end;
            }
            p = q;
        }
        rec.chunk.push_back(rec.end);
        rec.before.assign(rec.chunk.size(), 0);
        for (size_t i = 0; i + 1 < rec.chunk.size(); i++) pieces.push_back({r, i});
    }
    parallel_for(pieces.size(), threads, [&](size_t k) {
        Record &rec = recs[pieces[k].rec];
        rec.before[pieces[k].idx + 1] = strip_lines(rec.chunk[pieces[k].idx], rec.chunk[pieces[k].idx + 1], nullptr);
    });
    for (Record &rec : recs) {
        for (size_t i = 1; i < rec.before.size(); i++) rec.before[i] += rec.before[i - 1];
        rec.n_bases = rec.before.empty() ? 0 : rec.before.back();
    }
    return recs;
}

// the bases [lo, hi) of a record (hi <= n_bases) -> out: the pieces that hold them are stripped by `threads` threads.
// Returns the position of out[0] in the record (the start of the first piece touched, <= lo).
inline uint64_t materialize(const Record &rec, uint64_t lo, uint64_t hi, unsigned threads, std::vector<uint8_t> &out) {
    size_t c_lo = (size_t)(std::upper_bound(rec.before.begin(), rec.before.end(), lo) - rec.before.begin()) - 1;
    size_t c_hi = (size_t)(std::lower_bound(rec.before.begin(), rec.before.end(), hi) - rec.before.begin());
    if (c_hi > rec.chunk.size() - 1) c_hi = rec.chunk.size() - 1;
    if (c_hi < c_lo) c_hi = c_lo;
    const uint64_t buf_base = rec.before[c_lo];
    out.resize((size_t)(rec.before[c_hi] - buf_base));
    parallel_for(c_hi - c_lo, threads, [&](size_t k) {
        const size_t c = c_lo + k;
        strip_lines(rec.chunk[c], rec.chunk[c + 1], out.data() + (rec.before[c] - buf_base));
    });
    return buf_base;
}

// the same into a caller's buffer of `cap` bytes, on the calling thread (the native driver's workers strip one unit each,
// straight into pinned memory).  Returns the position of out[0] in the record, or ~0 if the pieces do not fit.
inline uint64_t materialize_into(const Record &rec, uint64_t lo, uint64_t hi, uint8_t *out, size_t cap) {
    size_t c_lo = (size_t)(std::upper_bound(rec.before.begin(), rec.before.end(), lo) - rec.before.begin()) - 1;
    size_t c_hi = (size_t)(std::lower_bound(rec.before.begin(), rec.before.end(), hi) - rec.before.begin());
    if (c_hi > rec.chunk.size() - 1) c_hi = rec.chunk.size() - 1;
    if (c_hi < c_lo) c_hi = c_lo;
    const uint64_t buf_base = rec.before[c_lo];
    if (rec.before[c_hi] - buf_base > cap) return ~0ULL;
    for (size_t c = c_lo; c < c_hi; c++) strip_lines(rec.chunk[c], rec.chunk[c + 1], out + (rec.before[c] - buf_base));
    return buf_base;
}

}  // namespace nm_fasta
#endif

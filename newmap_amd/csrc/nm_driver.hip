// nm_driver.hip -- native `search` driver: FASTA in -> <id>.unique.uintN files out, one call.
//
// SURVEY.md section 8(f) rank 1.  Once the kernels run at tens of G positions/s the Python loop of
// newmap/search.py:260-357 (read lines, slice segments, call, append) is what a user waits for.  This
// driver does that loop natively: a streaming FASTA reader with the record / segment rules of
// newmap/fasta.py:20-190 fills pinned host buffers; H2D copy, the fused kernels
// (nm_min_unique_segment_dev / nm_fixed_k_segment_dev) and the D2H copy of segment j run on the
// handle's stream while the host parses segment j+1 and appends segment j-1 to its file.
// Output files are byte-identical to newmap_amd.search.write_unique_counts (and so to the reference's).
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <zlib.h>

#include "../../include/newmap_amd.h"
#include "nm_internal.h"

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            nm_set_error("HIP error %d (%s) at %s:%d: %s", (int)e__, hipGetErrorString(e__),  \
                         __FILE__, __LINE__, #expr);                                          \
            return NM_E_DEVICE;                                                               \
        }                                                                                     \
    } while (0)

namespace {

inline bool is_space(unsigned char c) {      // what bytes.rstrip() removes (newmap/fasta.py:47)
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

struct Slot {                                 // one segment in flight
    uint8_t *h_in = nullptr;                  // pinned
    uint8_t *h_out = nullptr;                 // pinned
    uint64_t *h_status = nullptr;             // pinned, NM_STATUS_WORDS
    void *d_in = nullptr, *d_out = nullptr;
    uint64_t *d_status = nullptr;
    hipEvent_t done = nullptr;
    bool busy = false;
    uint64_t seg_len = 0, num_kmers = 0;
    uint64_t rec_index = 0;                   // which output file
    uint64_t rec_offset = 0;                  // first position of the segment inside its record
};

struct Driver {
    nm_index *ix;
    int device;
    hipStream_t stream;
    std::vector<uint32_t> ks;
    bool range_mode, use_rc;
    uint32_t kmin, kmax;
    int elem_bytes;
    uint64_t batch, lookahead;
    std::string out_dir, suffix;
    std::vector<std::string> include, exclude;
    nm_record_callback cb;
    void *user;
    Slot slots[2];
    int next_slot = 0;
    // current record
    std::string cur_id;
    bool have_record = false, record_wanted = false, any_processed = false;
    std::vector<uint8_t> buf;                 // record bytes not yet handed to the device
    uint64_t buf_offset = 0;                  // position of buf[0] in the record
    FILE *cur_file = nullptr;
    uint64_t file_serial = 0;                 // increments per opened output file
    std::vector<FILE *> open_files;           // indexed by file_serial - 1 (kept until drained)
    nm_search_summary rec_sum, total;
    std::vector<nm_search_summary> pending_sums;   // per file serial
    std::vector<std::string> pending_ids;
    std::vector<int> outstanding;             // segments in flight per file serial

    static void reset(nm_search_summary &s, uint32_t kmin, uint32_t kmax) {
        memset(&s, 0, sizeof s);
        s.max_len = kmin;                     // newmap/search.py:242-243
        s.min_len = kmax;
    }
};

bool wanted(const Driver &d, const std::string &id) {
    if (!d.include.empty()) {
        for (const auto &s : d.include) if (s == id) return true;
        return false;
    }
    for (const auto &s : d.exclude) if (s == id) return false;
    return true;
}

int alloc_slot(Driver &d, Slot &s) {
    const uint64_t in_bytes = d.batch + d.lookahead + 64;
    const uint64_t out_bytes = (d.batch + d.lookahead) * (uint64_t)d.elem_bytes + 64;
    HIP_TRY(hipHostMalloc((void **)&s.h_in, in_bytes, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&s.h_out, out_bytes, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&s.h_status, NM_STATUS_WORDS * sizeof(uint64_t), hipHostMallocDefault));
    HIP_TRY(hipMalloc(&s.d_in, in_bytes));
    HIP_TRY(hipMalloc(&s.d_out, out_bytes));
    HIP_TRY(hipMalloc((void **)&s.d_status, NM_STATUS_WORDS * sizeof(uint64_t)));
    HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    return NM_OK;
}

void free_slot(Slot &s) {
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_out) (void)hipHostFree(s.h_out);
    if (s.h_status) (void)hipHostFree(s.h_status);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.d_status) (void)hipFree(s.d_status);
    if (s.done) (void)hipEventDestroy(s.done);
    s = Slot();
}

// wait for a slot's segment, append its result to its file, fold its statistics
int drain_slot(Driver &d, Slot &s) {
    if (!s.busy) return NM_OK;
    HIP_TRY(hipEventSynchronize(s.done));
    s.busy = false;
    const uint64_t serial = s.rec_index;
    if (s.h_status[1]) {
        nm_set_error("a generated k-mer was not found in the index (record '%s', position %llu); possibly a "
                     "mismatch between the sequence and the index", d.pending_ids[serial].c_str(),
                     (unsigned long long)(s.rec_offset + s.h_status[2]));
        return NM_E_KMER_NOT_FOUND;
    }
    FILE *fp = d.open_files[serial];
    const uint64_t bytes = s.num_kmers * (uint64_t)d.elem_bytes;
    if (bytes && fwrite(s.h_out, 1, bytes, fp) != bytes) {
        nm_set_error("could not write the unique-length file of record '%s': %s", d.pending_ids[serial].c_str(), strerror(errno));
        return NM_E_FILE_WRITE;
    }
    // statistics of newmap/search.py:331-347
    nm_search_summary &rs = d.pending_sums[serial];
    // (branch-free reductions per element width: the compiler vectorises them; the branchy loop over every
    // element was the largest single host cost of a 3 Gbp run)
    uint64_t uniq = 0;
    uint32_t mx = 0, mn = 0xFFFFFFFFu;
    auto fold = [&](const auto *v, uint64_t n) {
        uint64_t u = 0;
        uint32_t hi = 0, lo = 0xFFFFFFFFu;
        for (uint64_t i = 0; i < n; i++) {
            const uint32_t x = v[i];
            u += x != 0;
            hi = x > hi ? x : hi;
            const uint32_t y = x ? x : 0xFFFFFFFFu;          // zeros do not take part in the minimum
            lo = y < lo ? y : lo;
        }
        uniq = u; mx = hi; mn = lo;
    };
    if (d.elem_bytes == 1) fold((const uint8_t *)s.h_out, s.num_kmers);
    else if (d.elem_bytes == 2) fold((const uint16_t *)s.h_out, s.num_kmers);
    else fold((const uint32_t *)s.h_out, s.num_kmers);
    rs.positions += s.num_kmers;
    rs.ambiguous += s.h_status[0];
    rs.unique += uniq;
    rs.no_unique += s.num_kmers - uniq - s.h_status[0];
    if (uniq) { if (mx > rs.max_len) rs.max_len = mx; if (mn < rs.min_len) rs.min_len = mn; }
    if (--d.outstanding[serial] == 0 && d.open_files[serial] != d.cur_file) {
        // the record is complete and no longer current: close and report it
        fclose(fp);
        d.open_files[serial] = nullptr;
    }
    return NM_OK;
}

// hand buf[0 .. seg_len) to the device as one segment with num_kmers positions
int submit(Driver &d, uint64_t seg_len, uint64_t num_kmers) {
    Slot &s = d.slots[d.next_slot];
    d.next_slot ^= 1;
    int rc = drain_slot(d, s);
    if (rc != NM_OK) return rc;
    memcpy(s.h_in, d.buf.data(), seg_len);
    s.seg_len = seg_len;
    s.num_kmers = num_kmers;
    s.rec_index = d.file_serial - 1;
    s.rec_offset = d.buf_offset;
    HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, seg_len, hipMemcpyHostToDevice, d.stream));
    if (d.range_mode)
        rc = nm_min_unique_segment_dev(d.ix, s.d_in, seg_len, num_kmers, d.kmin, d.kmax, d.use_rc, d.elem_bytes, s.d_out, s.d_status, d.stream);
    else
        rc = nm_fixed_k_segment_dev(d.ix, s.d_in, seg_len, num_kmers, d.ks.data(), (uint32_t)d.ks.size(), d.use_rc, d.elem_bytes, s.d_out, s.d_status, d.stream);
    if (rc != NM_OK) return rc;
    HIP_TRY(hipMemcpyAsync(s.h_out, s.d_out, num_kmers * (uint64_t)d.elem_bytes, hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipMemcpyAsync(s.h_status, s.d_status, NM_STATUS_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipEventRecord(s.done, d.stream));
    s.busy = true;
    d.outstanding[s.rec_index]++;
    return NM_OK;
}

// full segments while more than batch + lookahead bytes are buffered (newmap/fasta.py:109-150: a
// segment that ends exactly at the record end is the epilogue, so "==" waits for the end)
int pump(Driver &d) {
    while (d.buf.size() > d.batch + d.lookahead) {
        int rc = submit(d, d.batch + d.lookahead, d.batch);
        if (rc != NM_OK) return rc;
        d.buf.erase(d.buf.begin(), d.buf.begin() + (ptrdiff_t)d.batch);
        d.buf_offset += d.batch;
    }
    return NM_OK;
}

int end_record(Driver &d) {
    if (!d.have_record || !d.record_wanted) { d.buf.clear(); return NM_OK; }
    int rc = pump(d);
    if (rc != NM_OK) return rc;
    if (!d.buf.empty()) {
        rc = submit(d, d.buf.size(), d.buf.size());       // epilogue: every byte is a position
        if (rc != NM_OK) return rc;
    }
    d.buf.clear();
    return NM_OK;
}

// a record's first data byte arrived (newmap/search.py:268-305)
int begin_output(Driver &d) {
    if (d.cur_file && d.pending_ids.back() == d.cur_id) return NM_OK;   // same id as the previous record: keep appending
    const std::string path = d.out_dir + "/" + d.cur_id + ".unique." + d.suffix;
    FILE *fp = fopen(path.c_str(), "wb");                 // truncate on a new id (:304-305)
    if (!fp) { nm_set_error("could not open %s: %s", path.c_str(), strerror(errno)); return NM_E_FILE_WRITE; }
    // the previous file is closed once its last segment has drained
    if (d.cur_file) {
        const uint64_t prev = d.file_serial - 1;
        FILE *pf = d.cur_file;
        d.cur_file = nullptr;
        if (d.outstanding[prev] == 0) { fclose(pf); d.open_files[prev] = nullptr; }
    }
    d.cur_file = fp;
    d.open_files.push_back(fp);
    d.outstanding.push_back(0);
    d.pending_ids.push_back(d.cur_id);
    nm_search_summary s;
    Driver::reset(s, d.kmin, d.kmax);
    d.pending_sums.push_back(s);
    d.file_serial++;
    d.any_processed = true;
    return NM_OK;
}

int on_data(Driver &d, const unsigned char *p, size_t len) {
    if (!len) return NM_OK;
    if (!d.have_record) {                                 // data in front of any header: id ""
        d.have_record = true;
        d.cur_id.clear();
        d.record_wanted = wanted(d, d.cur_id);
        d.buf_offset = 0;
    }
    if (!d.record_wanted) return NM_OK;
    if (d.buf.empty() && d.buf_offset == 0) {
        int rc = begin_output(d);
        if (rc != NM_OK) return rc;
    }
    d.buf.insert(d.buf.end(), p, p + len);
    return pump(d);
}

int on_header(Driver &d, const unsigned char *p, size_t len) {
    int rc = end_record(d);
    if (rc != NM_OK) return rc;
    // id = first whitespace-delimited token minus its first byte (newmap/fasta.py:75)
    size_t e = 0;
    while (e < len && !is_space(p[e])) e++;
    d.cur_id.assign((const char *)p + 1, e ? e - 1 : 0);
    d.have_record = true;
    d.record_wanted = wanted(d, d.cur_id);
    d.buf_offset = 0;
    return NM_OK;
}

int on_line(Driver &d, const unsigned char *p, size_t len) {
    while (len && is_space(p[len - 1])) len--;
    if (len && (p[0] == '>' || p[0] == ';')) return on_header(d, p, len);
    return on_data(d, p, len);
}

int run(Driver &d, const char *fasta_path) {
    FILE *probe = fopen(fasta_path, "rb");
    if (!probe) { nm_set_error("could not open %s: %s", fasta_path, strerror(errno)); return NM_E_FILE_OPEN; }
    fclose(probe);
    gzFile gz = gzopen(fasta_path, "rb");                 // transparent for plain files (newmap/util.py:10-18)
    if (!gz) { nm_set_error("could not open %s", fasta_path); return NM_E_FILE_OPEN; }
    gzbuffer(gz, 1 << 20);
    std::vector<unsigned char> chunk(8 << 20), carry;
    int rc = NM_OK;
    for (;;) {
        const int got = gzread(gz, chunk.data(), (unsigned)chunk.size());
        if (got < 0) { nm_set_error("read error in %s", fasta_path); rc = NM_E_FILE_OPEN; break; }
        if (got == 0) break;
        size_t start = 0;
        const unsigned char *base = chunk.data();
        while (rc == NM_OK) {
            const unsigned char *nl = (const unsigned char *)memchr(base + start, '\n', (size_t)got - start);
            if (!nl) break;
            const size_t i = (size_t)(nl - base);
            if (!carry.empty()) {
                carry.insert(carry.end(), base + start, base + i);
                rc = on_line(d, carry.data(), carry.size());
                carry.clear();
            } else {
                rc = on_line(d, base + start, i - start);
            }
            start = i + 1;
        }
        if (rc != NM_OK) break;
        carry.insert(carry.end(), base + start, base + got);
    }
    if (rc == NM_OK && !carry.empty()) rc = on_line(d, carry.data(), carry.size());
    gzclose(gz);
    if (rc == NM_OK) rc = end_record(d);
    return rc;
}

}  // namespace

extern "C" int nm_search_fasta(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks,
                               uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                               const char *const *include_ids, uint32_t n_include,
                               const char *const *exclude_ids, uint32_t n_exclude,
                               nm_record_callback cb, void *user, nm_search_summary *total) {
    if (!ix || !fasta_path || !out_dir || !ks || nk == 0) { nm_set_error("null argument"); return NM_E_ARGUMENT; }
    if (batch == 0) { nm_set_error("batch must be positive"); return NM_E_ARGUMENT; }
    Driver d;
    d.ix = ix;
    d.device = (int)nm_index_info(ix, 10);
    d.ks.assign(ks, ks + nk);
    d.range_mode = range_mode != 0;
    d.use_rc = use_revcomp != 0;
    d.kmin = d.kmax = ks[0];
    for (uint32_t i = 1; i < nk; i++) { if (ks[i] < d.kmin) d.kmin = ks[i]; if (ks[i] > d.kmax) d.kmax = ks[i]; }
    if (d.kmin < 1) { nm_set_error("k-mer lengths must be >= 1"); return NM_E_ARGUMENT; }
    if (d.range_mode && d.kmin == d.kmax) { nm_set_error("math domain error: a k-mer range needs two different lengths"); return NM_E_ARGUMENT; }
    d.elem_bytes = d.kmax <= 0xFF ? 1 : (d.kmax <= 0xFFFF ? 2 : 4);      // newmap/search.py:204-212
    d.suffix = d.elem_bytes == 1 ? "uint8" : (d.elem_bytes == 2 ? "uint16" : "uint32");
    d.batch = batch;
    d.lookahead = d.kmax - 1;                                            // newmap/search.py:229
    d.out_dir = out_dir;
    for (uint32_t i = 0; i < n_include; i++) d.include.emplace_back(include_ids[i]);
    for (uint32_t i = 0; i < n_exclude; i++) d.exclude.emplace_back(exclude_ids[i]);
    d.cb = cb;
    d.user = user;
    Driver::reset(d.total, d.kmin, d.kmax);
    HIP_TRY(hipSetDevice(d.device));
    HIP_TRY(hipStreamCreate(&d.stream));
    int rc = NM_OK;
    for (auto &s : d.slots)
        if ((rc = alloc_slot(d, s)) != NM_OK) break;
    if (rc == NM_OK) rc = run(d, fasta_path);
    for (int i = 0; i < 2; i++) {                         // oldest segment first: appends stay in order
        Slot &s = d.slots[d.next_slot ^ i];
        if (rc == NM_OK) rc = drain_slot(d, s);
        else if (s.busy) (void)hipEventSynchronize(s.done);
    }
    for (FILE *&fp : d.open_files) if (fp) { fclose(fp); fp = nullptr; }
    // per-file summaries in file order, then the totals
    if (rc == NM_OK) {
        for (size_t i = 0; i < d.pending_sums.size(); i++) {
            nm_search_summary &rs = d.pending_sums[i];
            rs.records = 1;
            d.total.records++;
            d.total.positions += rs.positions;
            d.total.ambiguous += rs.ambiguous;
            d.total.unique += rs.unique;
            d.total.no_unique += rs.no_unique;
            if (rs.unique) {
                if (rs.max_len > d.total.max_len) d.total.max_len = rs.max_len;
                if (rs.min_len < d.total.min_len) d.total.min_len = rs.min_len;
            }
            if (cb) cb(d.pending_ids[i].c_str(), &rs, user);
        }
        if (total) *total = d.total;
        if (!d.any_processed) {
            nm_set_error(d.include.empty() ? "The excluded sequences were too strict and nothing was processed"
                                           : "None of the included sequences were found");
            if (!d.include.empty() || !d.exclude.empty()) rc = NM_E_ARGUMENT;
        }
    }
    for (auto &s : d.slots) free_slot(s);
    (void)hipStreamDestroy(d.stream);
    return rc;
}

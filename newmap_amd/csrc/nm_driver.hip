// nm_driver.hip -- native `search` driver: FASTA in -> <id>.unique.uintN files out, one call.
//
// SURVEY.md section 8(f) rank 1.  Once the kernels run at tens of G positions/s the Python loop of
// newmap/search.py:260-357 (read lines, slice segments, call, append) is what a user waits for.  This
// driver does that loop natively: a streaming FASTA reader with the record / segment rules of
// newmap/fasta.py:20-190 fills pinned host buffers; H2D copy, the fused kernels
// (nm_min_unique_segment_dev / nm_fixed_k_segment_dev) and the D2H copy of segment j run on the
// handle's stream while the host parses segment j+1 and appends segment j-1 to its file.
// Output files are byte-identical to newmap_amd.search.write_unique_counts (and so to the reference's).
//
// Two front-ends, one set of rules:
//   * the PARALLEL one (plain FASTA files; fast_run below): the file is mapped, header lines are found by a threaded
//     scan, every record's data lines are stripped by a pool of threads straight into one buffer, the segments go
//     through a ring of pinned slots (H2D, kernels, D2H on the driver's stream), and a pool of writer threads puts each
//     result into its file with pwrite while the next record is being stripped.  With world > 1 a rank strips and
//     searches only the byte ranges of its own work units (interleaved chunks of the position space) and writes them at
//     their offsets: no collective, no rank reads the whole file into memory.
//   * the STREAMING one (gzip input, anything that cannot be mapped; run below): one thread, two slots.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "../../include/newmap_amd.h"
#include "nm_fasta_scan.hpp"
#include "nm_hash.h"
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

using nm_fasta::is_space;

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
    int64_t rec_ordinal = -1;                 // which record (fingerprints)
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
    bool have_record = false, record_wanted = false, any_processed = false, new_record = false;
    std::vector<uint8_t> buf;                 // record bytes not yet handed to the device
    uint64_t buf_offset = 0;                  // position of buf[0] in the record
    FILE *cur_file = nullptr;
    uint64_t file_serial = 0;                 // increments per opened output file
    std::vector<FILE *> open_files;           // indexed by file_serial - 1 (kept until drained)
    nm_search_summary rec_sum, total;
    std::vector<nm_search_summary> pending_sums;   // per file serial
    std::vector<std::string> pending_ids;
    std::vector<int> outstanding;             // segments in flight per file serial
    // record fingerprints (nm_hash.h): ordinal = number of the record among the records with data, in file order
    int64_t rec_ordinal = -1;
    std::vector<uint64_t> rec_hash, rec_len;  // per ordinal: sum of the segments' fingerprints, bases
    std::vector<uint8_t> rec_searched;        // per ordinal: the record was searched (wanted)
    // guard pass (records that are not among the indexed ones): no files, nm_guard_segment_dev instead of the search
    bool guard_mode = false;
    const std::vector<uint8_t> *guard_set = nullptr;   // per ordinal: 1 = guard this record
    uint32_t initial_len = 0;

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
    HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming | hipEventBlockingSync));
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
        const uint64_t at = s.h_status[2] < s.seg_len ? s.h_status[2] : 0;
        const uint64_t len = s.seg_len - at < d.kmin ? s.seg_len - at : d.kmin;
        nm_set_error("The following generated k-mer was not found in the index:\n%.*s\nPossibly a mismatch between the sequence "
                     "and the index. (record '%s', position %llu)", (int)len, (const char *)s.h_in + at,      // newmap/search.py:719-722
                     d.guard_mode ? d.cur_id.c_str() : d.pending_ids[serial].c_str(), (unsigned long long)(s.rec_offset + at));
        return NM_E_KMER_NOT_FOUND;
    }
    if (d.guard_mode) return NM_OK;
    if (s.rec_ordinal >= 0) {                              // (a segment that does not start at a word of its record cannot be joined)
        if (s.rec_offset & 63u) d.rec_searched[(size_t)s.rec_ordinal] = 2;
        d.rec_hash[(size_t)s.rec_ordinal] += nm_hash_pow(s.rec_offset >> 6) * s.h_status[NM_STATUS_HASH];
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
    s.rec_index = d.guard_mode ? 0 : d.file_serial - 1;
    s.rec_offset = d.buf_offset;
    s.rec_ordinal = d.rec_ordinal;
    HIP_TRY(hipMemcpyAsync(s.d_in, s.h_in, seg_len, hipMemcpyHostToDevice, d.stream));
    if (d.guard_mode) {
        const uint32_t two[2] = {d.kmin, d.kmax};
        rc = nm_guard_segment_dev(d.ix, s.d_in, seg_len, num_kmers, d.range_mode ? two : d.ks.data(), d.range_mode ? 2u : (uint32_t)d.ks.size(),
                                  d.range_mode, d.initial_len, d.use_rc, s.d_status, d.stream);
        if (rc != NM_OK) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_status, s.d_status, NM_STATUS_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, d.stream));
        HIP_TRY(hipEventRecord(s.done, d.stream));
        s.busy = true;
        return NM_OK;
    }
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
    if (d.guard_mode) return NM_OK;
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
        d.new_record = true;
    }
    if (d.new_record) {                                   // the record's first data byte: it counts (records without data do not)
        d.new_record = false;
        d.rec_ordinal++;
        if (!d.guard_mode) { d.rec_hash.push_back(0); d.rec_len.push_back(0); d.rec_searched.push_back(d.record_wanted ? 1 : 0); }
        else d.record_wanted = d.record_wanted && (size_t)d.rec_ordinal < d.guard_set->size() && (*d.guard_set)[(size_t)d.rec_ordinal];
    }
    if (!d.guard_mode) d.rec_len[(size_t)d.rec_ordinal] += len;
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
    d.new_record = true;
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


// ------------------------------------------------------------------------------ parallel front-end

using nm_fasta::parallel_for;
using nm_fasta::host_threads;
typedef nm_fasta::Record FastRecord;

struct FastFile {
    std::string id, path;
    int fd = -1;
    uint64_t n_elems = 0;
    nm_search_summary sum;
    std::atomic<long> outstanding{0};
    std::atomic<bool> submitted_all{false};                // every unit of the file has been handed to the device
    int last_record = -1;                                  // the last record that writes into it
    bool reported = false;
};

struct FastSlot {
    uint8_t *h_in = nullptr, *h_out = nullptr;
    uint64_t *h_status = nullptr, *h_sum = nullptr;        // pinned: NM_STATUS_WORDS, 3 (non-zero elements, largest, smallest non-zero)
    void *d_in = nullptr, *d_out = nullptr;
    uint64_t *d_status = nullptr, *d_sum = nullptr;
    hipEvent_t done = nullptr;
    uint64_t seg_len = 0, count = 0, rec_start = 0;
    int rec = -1;
};

// non-zero elements, the largest and the smallest non-zero one (newmap/search.py:331-347), on the device
#ifdef NM_DRIVER_HOST_SUMMARY      /* tests/test_sanitizers.py: the driver's host logic under ThreadSanitizer, "device" = host memory */
template <typename T>
void k_out_summary(const T *out, uint64_t n, unsigned long long *sum) {
    unsigned long long cnt = 0, mx = 0, mn = ~0ULL;
    for (uint64_t i = 0; i < n; i++) { const unsigned long long v = out[i]; if (v) { cnt++; mx = v > mx ? v : mx; mn = v < mn ? v : mn; } }
    if (cnt) { sum[0] += cnt; sum[1] = mx > sum[1] ? mx : sum[1]; sum[2] = mn < sum[2] ? mn : sum[2]; }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) kernel(__VA_ARGS__)
#else
template <typename T>
__global__ void k_out_summary(const T *__restrict__ out, uint64_t n, unsigned long long *__restrict__ sum) {
    unsigned long long cnt = 0, mx = 0, mn = ~0ULL;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long v = out[i];
        if (v) { cnt++; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        const unsigned long long a = __shfl_down(mx, off, 64), b = __shfl_down(mn, off, 64);
        mx = a > mx ? a : mx; mn = b < mn ? b : mn;
    }
    if ((threadIdx.x & 63) == 0 && cnt) {
        atomicAdd(&sum[0], cnt);
        atomicMax(&sum[1], mx);
        atomicMin(&sum[2], mn);
    }
}
#endif

struct FastDriver {
    nm_index *ix = nullptr;
    int device = 0;
    std::vector<hipStream_t> streams;     // worker i runs on streams[i % n]: copies and kernels of neighbouring units overlap (lanes of the handle)
    std::vector<uint32_t> ks;
    bool range_mode = true, use_rc = true;
    uint32_t kmin = 0, kmax = 0;
    int elem_bytes = 1;
    uint64_t batch = 0, lookahead = 0;
    std::vector<FastRecord> recs;
    std::deque<FastFile> files;
    std::vector<FastSlot> slots;
    std::atomic<int> error{NM_OK};
    std::string error_text;
    std::mutex sum_mu;
    std::vector<uint64_t> rec_hash;       // per record of `recs`: sum of its segments' fingerprints (nm_hash.h), under sum_mu
    std::vector<uint8_t> rec_unaligned;   // ... a segment of it did not start at a multiple of 64: the sum means nothing

    void fail(int code, const std::string &text) {
        int expect = NM_OK;
        if (error.compare_exchange_strong(expect, code)) { std::lock_guard<std::mutex> g(sum_mu); error_text = text; }
    }
};

// returns NM_OK, an error, or -1 when this front-end does not apply (gzip input, a file that cannot be mapped)
// rec_info (may be null): per record WITH data, in file order, {length, fingerprint summed over THIS rank's units, 1 if the
// record was searched}; *n_rec_info = their number (filled up to rec_cap records)
int fast_run(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks, uint32_t nk, int range_mode,
             int use_revcomp, uint64_t batch, const std::vector<std::string> &include, const std::vector<std::string> &exclude,
             nm_record_callback cb, void *user, nm_search_summary *total, int rank, int world,
             std::vector<uint64_t> *rec_info) {
    const int fd = open(fasta_path, O_RDONLY);
    if (fd < 0) { nm_set_error("could not open %s: %s", fasta_path, strerror(errno)); return NM_E_FILE_OPEN; }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return -1; }
    size_t size = (size_t)st.st_size;
    unsigned char magic[2] = {0, 0};
    const unsigned char *base = nullptr;
    unsigned char *inflated = nullptr;                        // gzip input (newmap/util.py:10-18): inflated once into memory, then the same front-end
    if (size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        close(fd);
        gzFile gz = gzopen(fasta_path, "rb");
        if (!gz) return -1;
        (void)gzbuffer(gz, 1u << 20);
        size_t cap = size * 4 + (1u << 20), used = 0;
        inflated = (unsigned char *)malloc(cap);
        while (inflated) {
            if (cap - used < (1u << 24)) {
                cap += cap / 2 + (1u << 24);
                unsigned char *grown = (unsigned char *)realloc(inflated, cap);
                if (!grown) { free(inflated); inflated = nullptr; break; }
                inflated = grown;
            }
            const size_t want = cap - used < (1u << 30) ? cap - used : (size_t)(1u << 30);
            const int got = gzread(gz, inflated + used, (unsigned)want);
            if (got < 0) { free(inflated); inflated = nullptr; gzclose(gz); nm_set_error("could not inflate %s", fasta_path); return NM_E_FILE_OPEN; }
            if (got == 0) break;
            used += (size_t)got;
        }
        gzclose(gz);
        if (!inflated) { nm_set_error("out of memory inflating %s", fasta_path); return NM_E_ALLOC; }
        base = inflated;
        size = used;
    } else {
        if (size) {
            void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); return -1; }
            (void)madvise(m, size, MADV_WILLNEED);
            base = (const unsigned char *)m;
        }
        close(fd);
    }
    const unsigned threads = host_threads();
    FastDriver d;
    d.ix = ix;
    d.device = (int)nm_index_info(ix, 10);
    d.ks.assign(ks, ks + nk);
    d.range_mode = range_mode != 0;
    d.use_rc = use_revcomp != 0;
    d.kmin = d.kmax = ks[0];
    for (uint32_t i = 1; i < nk; i++) { if (ks[i] < d.kmin) d.kmin = ks[i]; if (ks[i] > d.kmax) d.kmax = ks[i]; }
    d.elem_bytes = d.kmax <= 0xFF ? 1 : (d.kmax <= 0xFFFF ? 2 : 4);      // newmap/search.py:204-212
    const char *suffix = d.elem_bytes == 1 ? "uint8" : (d.elem_bytes == 2 ? "uint16" : "uint32");
    d.batch = batch >= 64 ? batch & ~63ull : batch;          // (segments start at multiples of 64 bases of their record: their fingerprints join, nm_hash.h)
    d.lookahead = d.kmax - 1;                                            // newmap/search.py:229
    auto unmap = [&]() { if (inflated) free(inflated); else if (base) munmap((void *)base, size); };

    // ---- records, their pieces, the bases in front of each piece (nm_fasta_scan.hpp; threaded)
    d.recs = nm_fasta::scan(base, size, threads);
    d.rec_hash.assign(d.recs.size(), 0);
    d.rec_unaligned.assign(d.recs.size(), 0);
    // ---- output files: one per run of adjacent records (that hold data and are wanted) with one id; an id that
    // comes back later truncates the file again (newmap/search.py:268-305), so only its LAST run is searched
    auto wanted = [&](const std::string &id) {
        if (!include.empty()) { for (const auto &s : include) if (s == id) return true; return false; }
        for (const auto &s : exclude) if (s == id) return false;
        return true;
    };
    {
        int cur = -1;
        std::string cur_id;
        for (FastRecord &rec : d.recs) {
            if (rec.n_bases == 0) continue;                     // a record without data yields nothing (newmap/fasta.py:173-188)
            if (!wanted(rec.id)) continue;                      // (a skipped record does not end the current run: newmap/search.py:268-305
                                                                //  never updates its current id on one; same as the streaming front-end)
            if (cur < 0 || rec.id != cur_id) {
                d.files.emplace_back();
                cur = (int)d.files.size() - 1;
                cur_id = rec.id;
                d.files[cur].id = rec.id;
                d.files[cur].path = std::string(out_dir) + "/" + rec.id + ".unique." + suffix;
                memset(&d.files[cur].sum, 0, sizeof(nm_search_summary));
                d.files[cur].sum.max_len = d.kmin;               // newmap/search.py:242-243
                d.files[cur].sum.min_len = d.kmax;
            }
            rec.file = cur;
            rec.file_offset = d.files[cur].n_elems;
            d.files[cur].n_elems += rec.n_bases;
        }
        {   // a run whose id comes back in a later run is superseded (one pass over the ids)
            std::unordered_map<std::string, size_t> last_run;
            for (size_t a = 0; a < d.files.size(); a++) last_run[d.files[a].id] = a;
            for (size_t a = 0; a < d.files.size(); a++) if (last_run[d.files[a].id] != a) d.files[a].n_elems = 0;
        }
        for (FastRecord &rec : d.recs) if (rec.file >= 0 && d.files[rec.file].n_elems == 0) rec.file = -1;
        for (size_t r = 0; r < d.recs.size(); r++) if (d.recs[r].file >= 0) d.files[d.recs[r].file].last_record = (int)r;
    }
    uint64_t total_positions = 0;
    for (FastRecord &rec : d.recs) if (rec.file >= 0) { rec.global = total_positions; total_positions += rec.n_bases; }
    bool any = false;
    for (const FastFile &f : d.files) any = any || f.n_elems;
    int rc = NM_OK;
    if (!any) {
        unmap();
        if (total) { memset(total, 0, sizeof *total); total->max_len = d.kmin; total->min_len = d.kmax; }
        if (!include.empty() || !exclude.empty()) {
            nm_set_error(include.empty() ? "The excluded sequences were too strict and nothing was processed" : "None of the included sequences were found");
            return NM_E_ARGUMENT;
        }
        return NM_OK;
    }
    for (FastFile &f : d.files) {
        if (!f.n_elems) continue;
        // (every rank sets the same length: the ranks write disjoint ranges of one file)
        f.fd = open(f.path.c_str(), O_WRONLY | O_CREAT | (world == 1 ? O_TRUNC : 0), 0666);
        if (f.fd < 0 || ftruncate(f.fd, (off_t)(f.n_elems * (uint64_t)d.elem_bytes)) != 0) {
            nm_set_error("could not open %s: %s", f.path.c_str(), strerror(errno));
            rc = NM_E_FILE_WRITE;
            break;
        }
    }
    // ---- this rank's ranges of the position space: interleaved chunks of ~64 M positions (newmap_amd/parallel.py)
    std::vector<std::pair<uint64_t, uint64_t>> ranges;
    if (world <= 1) ranges.push_back({0, total_positions});
    else {
        uint64_t target = 64ull << 20;
        if (const char *e = getenv("NEWMAP_AMD_SHARD_CHUNK")) { const long long v = atoll(e); if (v > 0) target = (uint64_t)v; }   // (tests)
        const uint64_t rounds = (total_positions + (uint64_t)world * target - 1) / ((uint64_t)world * target);
        const uint64_t n_chunks = (uint64_t)world * (rounds ? rounds : 1);
        for (uint64_t c = (uint64_t)rank; c < n_chunks; c += (uint64_t)world) {
            const uint64_t lo = (uint64_t)((unsigned __int128)total_positions * c / n_chunks), hi = (uint64_t)((unsigned __int128)total_positions * (c + 1) / n_chunks);
            if (hi > lo) ranges.push_back({lo, hi});
        }
    }
    // ---- the units of my ranges, in file order.  A unit = as many whole batches as fit 32 M positions: the batch bounds what
    // the REFERENCE keeps in host memory per segment (newmap/main.py:172-180); here it only sets where segments may be cut,
    // the output does not depend on it, and a unit of 10 M positions spends as long in copies and launches as in its kernels.
    // NEWMAP_AMD_DRIVER_FUSE=0: one batch per unit.
    {
        const char *fz = getenv("NEWMAP_AMD_DRIVER_FUSE");
        const uint64_t launch = 32ull << 20;
        if (!(fz && fz[0] == '0') && d.batch < launch) d.batch *= launch / d.batch;
        // (units after a record's first start at a multiple of the working batch: kept a multiple of 64 bases, so that their
        // fingerprints join -- nm_hash.h -- whatever --kmer-batch-size is; the output does not depend on where segments are cut)
        if (d.batch >= 64) d.batch &= ~63ull;
    }
    struct Unit { int rec; uint64_t start, count, seg_len; };
    std::vector<Unit> units;
    std::vector<long> file_units(d.files.size(), 0);
    for (size_t ri = 0; ri < d.recs.size() && rc == NM_OK; ri++) {
        FastRecord &rec = d.recs[ri];
        if (rec.file < 0) continue;
        for (auto &rg : ranges) {
            const uint64_t lo = rg.first > rec.global ? rg.first : rec.global;
            const uint64_t hi = rg.second < rec.global + rec.n_bases ? rg.second : rec.global + rec.n_bases;
            // (a part starts and ends at a multiple of 64 bases of its record -- or at the record's end --, so that the segments'
            // fingerprints can be joined (nm_hash.h); every rank rounds a shared boundary the same way)
            auto word_edge = [&](uint64_t x) { return x == rec.n_bases ? x : x & ~63ull; };
            if (hi <= lo) continue;
            const uint64_t first = word_edge(lo - rec.global), last = word_edge(hi - rec.global);
            for (uint64_t p = first; p < last; p += d.batch) {
                const uint64_t count = last - p < d.batch ? last - p : d.batch;
                units.push_back({(int)ri, p, count, (p + count + d.lookahead < rec.n_bases ? p + count + d.lookahead : rec.n_bases) - p});
                file_units[rec.file]++;
            }
        }
    }
    // ---- workers.  Every worker owns one slot (pinned input / output, device buffers, an event) and takes the units in
    // turn: it strips the unit's lines straight into its pinned buffer, submits copy-in, kernels and copy-out (one
    // worker at a time: the handle's calls are made one by one), waits for its event and pwrite()s the result into the
    // record's file.  Strip, transfers, kernels and file writes of different units overlap; nothing is staged twice.
    int n_workers = d.batch <= (40u << 20) ? 10 : 4;          // (measured on 3.09 Gbp, 16 CPUs granted: 10 workers of 32 M units on 5 streams; 16 workers contend)
    if (const char *e = getenv("NEWMAP_AMD_DRIVER_SLOTS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n_workers = v; }
    if ((size_t)n_workers > units.size()) n_workers = units.empty() ? 1 : (int)units.size();
    const size_t piece_slack = 2 * nm_fasta::kPieceBytes + 4096;    // a unit is stripped piece-wise: whole pieces around it
    const uint64_t in_bytes = d.batch + d.lookahead + 64, out_bytes = d.batch * (uint64_t)d.elem_bytes + 64;
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        d.fail(NM_E_DEVICE, std::string("HIP error (") + hipGetErrorString(e) + "): " + what);
        return false;
    };
    // (each stream is a chain copy-in -> kernels -> copy-out: a unit of 32 M positions holds its stream for ~4 ms, most of it the
    // two DMA copies, so the number of streams is the number of copies in flight -- 3.09 Gbp: 2 streams 0.33 s, 4 streams 0.26 s,
    // 5 streams 0.244 s)
    int n_streams = 5;                                          // NEWMAP_AMD_DRIVER_STREAMS = 1 .. 5 (the handle has five lanes for caller streams)
    if (const char *e = getenv("NEWMAP_AMD_DRIVER_STREAMS")) { const int v = atoi(e); if (v >= 1 && v <= 5) n_streams = v; }
    const bool phase_times = getenv("NEWMAP_AMD_DRIVER_TIMING") != nullptr;
    bool zero_copy = false;
    if (const char *e = getenv("NEWMAP_AMD_DRIVER_ZEROCOPY")) zero_copy = e[0] == '1';
    std::atomic<uint64_t> t_strip_us{0}, t_submit_us{0}, t_wait_us{0}, t_write_us{0};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    bool streams_ok = rc == NM_OK && hip_ok(hipSetDevice(d.device), "hipSetDevice");
    for (int i = 0; i < n_streams && streams_ok; i++) {
        hipStream_t st = nullptr;
        streams_ok = hip_ok(hipStreamCreate(&st), "hipStreamCreate");
        if (streams_ok) d.streams.push_back(st);
    }
    // (every worker allocates its own slot as it starts -- 130 MB of pinned memory each: ten of them one after the other
    // held the first strip back by 40 - 60 ms)
    if (streams_ok) d.slots.resize(n_workers);
    auto make_slot = [&](FastSlot &s) {
        if (!hip_ok(hipHostMalloc((void **)&s.h_in, in_bytes + piece_slack, hipHostMallocDefault), "pinned input") ||
            !hip_ok(hipHostMalloc((void **)&s.h_out, out_bytes, hipHostMallocDefault), "pinned output") ||
            !hip_ok(hipHostMalloc((void **)&s.h_status, (NM_STATUS_WORDS + 3) * sizeof(uint64_t), hipHostMallocDefault), "pinned status") ||
            !hip_ok(hipMalloc(&s.d_in, in_bytes), "device input") || !hip_ok(hipMalloc(&s.d_out, out_bytes), "device output") ||
            !hip_ok(hipMalloc((void **)&s.d_status, (NM_STATUS_WORDS + 3) * sizeof(uint64_t)), "device status") ||
            !hip_ok(hipEventCreateWithFlags(&s.done, hipEventDisableTiming | hipEventBlockingSync), "event")) return false;
        s.h_sum = s.h_status + NM_STATUS_WORDS;
        s.d_sum = s.d_status + NM_STATUS_WORDS;
        return true;
    };
    if (rc == NM_OK && d.error.load() != NM_OK) rc = d.error.load();
    const double t_setup = now() - t_begin;
    std::atomic<size_t> next_unit{0};
    std::vector<std::atomic<long>> file_done(d.files.size());
    for (auto &x : file_done) x = 0;
    std::mutex submit_mu, done_mu;
    std::condition_variable done_cv;
    size_t units_done = 0;
    auto worker = [&](int si) {
        (void)hipSetDevice(d.device);
        FastSlot &s = d.slots[si];
        const bool slot_ok = make_slot(s);                 // (a failure is in d.error: the loop below then only counts the units off)
        (void)slot_ok;
        hipStream_t st = d.streams[(size_t)si % d.streams.size()];
        for (size_t ui; (ui = next_unit.fetch_add(1)) < units.size();) {
            const Unit &u = units[ui];
            const FastRecord &r = d.recs[u.rec];
            FastFile &f = d.files[r.file];
            if (d.error.load() == NM_OK) {
                double t0 = now();
                uint64_t buf_base = nm_fasta::materialize_into(r, u.start, u.start + u.seg_len, s.h_in, in_bytes + piece_slack);
                if (buf_base == ~0ULL) {                              // (lines far longer than a piece: strip aside, keep what is needed)
                    std::vector<uint8_t> tmp;
                    const uint64_t tb = nm_fasta::materialize(r, u.start, u.start + u.seg_len, 1, tmp);
                    memcpy(s.h_in, tmp.data() + (u.start - tb), u.seg_len);
                    buf_base = u.start;
                }
                const uint8_t *src = s.h_in + (u.start - buf_base);
                t_strip_us += (uint64_t)((now() - t0) * 1e6);
                s.rec = u.rec; s.rec_start = u.start; s.count = u.count; s.seg_len = u.seg_len;
                t0 = now();
                bool ok;
                {
                    std::lock_guard<std::mutex> g(submit_mu);
                    // (NEWMAP_AMD_DRIVER_ZEROCOPY=1, experiment: the kernels read the pinned slot over the bus themselves)
                    const void *dev_in = zero_copy ? (const void *)src : (const void *)s.d_in;
                    ok = zero_copy || hip_ok(hipMemcpyAsync(s.d_in, src, u.seg_len, hipMemcpyHostToDevice, st), "copy to device");
                    if (ok) {
                        const int e = d.range_mode
                            ? nm_min_unique_segment_dev(d.ix, dev_in, u.seg_len, u.count, d.kmin, d.kmax, d.use_rc, d.elem_bytes, s.d_out, s.d_status, st)
                            : nm_fixed_k_segment_dev(d.ix, dev_in, u.seg_len, u.count, d.ks.data(), (uint32_t)d.ks.size(), d.use_rc, d.elem_bytes, s.d_out, s.d_status, st);
                        if (e != NM_OK) { d.fail(e, nm_last_error()); ok = false; }
                    }
                    if (ok) {
                        ok = hip_ok(hipMemsetAsync(s.d_sum, 0, 16, st), "summary reset") &&             // count, largest
                             hip_ok(hipMemsetAsync(s.d_sum + 2, 0xFF, 8, st), "summary reset");        // smallest non-zero
                        const unsigned grid = (unsigned)((u.count + 256 * 16 - 1) / (256 * 16) < 2048 ? (u.count + 256 * 16 - 1) / (256 * 16) : 2048);
                        if (d.elem_bytes == 1) hipLaunchKernelGGL(k_out_summary<uint8_t>, dim3(grid ? grid : 1), dim3(256), 0, st, (const uint8_t *)s.d_out, u.count, (unsigned long long *)s.d_sum);
                        else if (d.elem_bytes == 2) hipLaunchKernelGGL(k_out_summary<uint16_t>, dim3(grid ? grid : 1), dim3(256), 0, st, (const uint16_t *)s.d_out, u.count, (unsigned long long *)s.d_sum);
                        else hipLaunchKernelGGL(k_out_summary<uint32_t>, dim3(grid ? grid : 1), dim3(256), 0, st, (const uint32_t *)s.d_out, u.count, (unsigned long long *)s.d_sum);
                        ok = ok && hip_ok(hipMemcpyAsync(s.h_out, s.d_out, u.count * (uint64_t)d.elem_bytes, hipMemcpyDeviceToHost, st), "copy from device") &&
                             hip_ok(hipMemcpyAsync(s.h_status, s.d_status, (NM_STATUS_WORDS + 3) * sizeof(uint64_t), hipMemcpyDeviceToHost, st), "status copy") &&
                             hip_ok(hipEventRecord(s.done, st), "event record");
                    }
                }
                t_submit_us += (uint64_t)((now() - t0) * 1e6);
                t0 = now();
                if (ok && hipEventSynchronize(s.done) != hipSuccess) { d.fail(NM_E_DEVICE, "waiting for a segment failed"); ok = false; }
                t_wait_us += (uint64_t)((now() - t0) * 1e6);
                t0 = now();
                if (ok && d.error.load() == NM_OK) {
                    if (s.h_status[1]) {
                        const uint64_t at = s.h_status[2] < s.seg_len ? s.h_status[2] : 0;
                        const uint64_t len = s.seg_len - at < d.kmin ? s.seg_len - at : d.kmin;
                        char buf[1024];
                        snprintf(buf, sizeof buf, "The following generated k-mer was not found in the index:\n%.*s\nPossibly a mismatch between the "
                                 "sequence and the index. (record '%s', position %llu)", (int)len, (const char *)src + at,   // newmap/search.py:719-722
                                 r.id.c_str(), (unsigned long long)(s.rec_start + at));
                        d.fail(NM_E_KMER_NOT_FOUND, buf);
                    } else {
                        const uint64_t bytes = s.count * (uint64_t)d.elem_bytes;
                        uint64_t off = (r.file_offset + s.rec_start) * (uint64_t)d.elem_bytes, done = 0;
                        while (done < bytes) {
                            const ssize_t w = pwrite(f.fd, s.h_out + done, bytes - done, (off_t)(off + done));
                            if (w < 0) { if (errno == EINTR) continue; d.fail(NM_E_FILE_WRITE, "could not write " + f.path + ": " + strerror(errno)); break; }
                            done += (uint64_t)w;
                        }
                        std::lock_guard<std::mutex> g(d.sum_mu);
                        d.rec_hash[(size_t)s.rec] += nm_hash_pow(s.rec_start >> 6) * s.h_status[NM_STATUS_HASH];
                        if (s.rec_start & 63u) d.rec_unaligned[(size_t)s.rec] = 1;   // (cannot be joined: the record goes to the guard)
                        nm_search_summary &rs = f.sum;                 // newmap/search.py:331-347
                        const uint64_t uniq = s.h_sum[0];
                        rs.positions += s.count;
                        rs.ambiguous += s.h_status[0];
                        rs.unique += uniq;
                        rs.no_unique += s.count - uniq - s.h_status[0];
                        if (uniq) {
                            if ((uint32_t)s.h_sum[1] > rs.max_len) rs.max_len = (uint32_t)s.h_sum[1];
                            if ((uint32_t)s.h_sum[2] < rs.min_len) rs.min_len = (uint32_t)s.h_sum[2];
                        }
                    }
                }
                t_write_us += (uint64_t)((now() - t0) * 1e6);
            }
            file_done[r.file]++;
            { std::lock_guard<std::mutex> g(done_mu); units_done++; }
            done_cv.notify_one();
        }
    };
    size_t next_report = 0;
    auto report_ready = [&]() {                                 // per-file summaries, in file order, as soon as a file has drained
        while (next_report < d.files.size()) {
            FastFile &f = d.files[next_report];
            if (f.n_elems) {
                if (file_done[next_report].load() != file_units[next_report]) break;
                f.sum.records = 1;
                if (cb && d.error.load() == NM_OK && (world <= 1 || file_units[next_report])) cb(f.id.c_str(), &f.sum, user);
                f.reported = true;
            }
            next_report++;
        }
    };
    std::vector<std::thread> workers;
    if (rc == NM_OK) {
        for (int i = 0; i < n_workers; i++) workers.emplace_back(worker, i);
        std::unique_lock<std::mutex> lk(done_mu);
        while (units_done < units.size()) {
            done_cv.wait(lk);
            lk.unlock();
            report_ready();
            lk.lock();
        }
    }
    for (auto &w : workers) w.join();
    if (rc == NM_OK && d.error.load() != NM_OK) { rc = d.error.load(); nm_set_error("%s", d.error_text.c_str()); }
    for (FastFile &f : d.files) if (f.fd >= 0) { if (close(f.fd) != 0 && rc == NM_OK) { nm_set_error("could not close %s: %s", f.path.c_str(), strerror(errno)); rc = NM_E_FILE_WRITE; } f.fd = -1; }
    for (FastSlot &s : d.slots) {
        if (s.h_in) (void)hipHostFree(s.h_in);
        if (s.h_out) (void)hipHostFree(s.h_out);
        if (s.h_status) (void)hipHostFree(s.h_status);
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.d_status) (void)hipFree(s.d_status);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (hipStream_t st : d.streams) { (void)nm_stream_release(ix, st); (void)hipStreamDestroy(st); }
    unmap();
    if (phase_times)
        fprintf(stderr, "[driver] total %.3fs (streams, pinned slots, device buffers: %.3f), %d workers, %zu units; summed over the workers: strip %.3f, submit %.3f, wait for the device %.3f, write %.3f s\n",
                now() - t_begin, t_setup, n_workers, units.size(), t_strip_us.load() * 1e-6, t_submit_us.load() * 1e-6, t_wait_us.load() * 1e-6, t_write_us.load() * 1e-6);
    if (rc != NM_OK) return rc;
    if (rec_info) {
        rec_info->clear();
        for (size_t ri = 0; ri < d.recs.size(); ri++) {
            if (d.recs[ri].n_bases == 0) continue;
            rec_info->push_back(d.recs[ri].n_bases);
            rec_info->push_back(d.rec_hash[ri]);
            rec_info->push_back(d.recs[ri].file >= 0 ? (d.rec_unaligned[ri] ? 2 : 1) : 0);
        }
    }
    report_ready();
    // ---- summaries in file order (this rank's share when world > 1), then the totals
    nm_search_summary tot;
    memset(&tot, 0, sizeof tot);
    tot.max_len = d.kmin;
    tot.min_len = d.kmax;
    for (size_t fi = 0; fi < d.files.size(); fi++) {
        FastFile &f = d.files[fi];
        if (!f.n_elems) continue;
        nm_search_summary &rs = f.sum;
        rs.records = 1;
        tot.records++;
        tot.positions += rs.positions;
        tot.ambiguous += rs.ambiguous;
        tot.unique += rs.unique;
        tot.no_unique += rs.no_unique;
        if (rs.unique) {
            if (rs.max_len > tot.max_len) tot.max_len = rs.max_len;
            if (rs.min_len < tot.min_len) tot.min_len = rs.min_len;
        }
        if (cb && !f.reported && (world <= 1 || file_units[fi])) cb(f.id.c_str(), &rs, user);
    }
    if (total) *total = tot;
    return NM_OK;
}

}  // namespace

// The exact guard over the records flagged in `flags` (per record with data, in file order): the streaming reader with
// nm_guard_segment_dev in the place of the search, nothing written.  NM_E_KMER_NOT_FOUND with the reference's message
// when one of its probes is absent (newmap/search.py:699-722).
static int guard_pass(nm_index *ix, const char *fasta_path, const uint32_t *ks, uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                      const std::vector<std::string> &include, const std::vector<std::string> &exclude, const std::vector<uint8_t> &flags) {
    Driver d;
    d.ix = ix;
    d.device = (int)nm_index_info(ix, 10);
    d.ks.assign(ks, ks + nk);
    d.range_mode = range_mode != 0;
    d.use_rc = use_revcomp != 0;
    d.kmin = d.kmax = ks[0];
    for (uint32_t i = 1; i < nk; i++) { if (ks[i] < d.kmin) d.kmin = ks[i]; if (ks[i] > d.kmax) d.kmax = ks[i]; }
    d.elem_bytes = 1;
    d.batch = batch >= 64 ? batch & ~63ull : batch;          // (segments start at multiples of 64 bases of their record: their fingerprints join, nm_hash.h)
    d.lookahead = d.kmax - 1;
    d.include = include;
    d.exclude = exclude;
    d.cb = nullptr;
    d.user = nullptr;
    d.guard_mode = true;
    d.guard_set = &flags;
    d.initial_len = (uint32_t)nm_index_info(ix, 22);
    Driver::reset(d.total, d.kmin, d.kmax);
    HIP_TRY(hipSetDevice(d.device));
    HIP_TRY(hipStreamCreate(&d.stream));
    int rc = NM_OK;
    for (auto &s : d.slots)
        if ((rc = alloc_slot(d, s)) != NM_OK) break;
    if (rc == NM_OK) rc = run(d, fasta_path);
    for (int i = 0; i < 2; i++) {
        Slot &s = d.slots[d.next_slot ^ i];
        if (rc == NM_OK) rc = drain_slot(d, s);
        else if (s.busy) (void)hipEventSynchronize(s.done);
    }
    for (auto &s : d.slots) free_slot(s);
    (void)nm_stream_release(ix, d.stream);
    (void)hipStreamDestroy(d.stream);
    return rc;
}

// which of the searched records are NOT among the indexed ones (info = {length, fingerprint, searched} per record)
static bool unverified_records(const nm_index *ix, const std::vector<uint64_t> &info, std::vector<uint8_t> &flags) {
    bool any = false;
    flags.assign(info.size() / 3, 0);
    for (size_t i = 0; i + 2 < info.size(); i += 3)
        if (info[i + 2] && (info[i + 2] == 2 || !nm_index_has_record(ix, info[i], info[i + 1]))) { flags[i / 3] = 1; any = true; }   // (2: not joinable)
    return any;
}

static int search_fasta_impl(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks,
                             uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                             const char *const *include_ids, uint32_t n_include,
                             const char *const *exclude_ids, uint32_t n_exclude,
                             nm_record_callback cb, void *user, nm_search_summary *total, int rank, int world,
                             std::vector<uint64_t> *rec_info_out = nullptr) {
    if (!ix || !fasta_path || !out_dir || !ks || nk == 0) { nm_set_error("null argument"); return NM_E_ARGUMENT; }
    if (batch == 0) { nm_set_error("batch must be positive"); return NM_E_ARGUMENT; }
    if (world < 1 || rank < 0 || rank >= world) { nm_set_error("rank %d of %d", rank, world); return NM_E_ARGUMENT; }
    uint32_t kmin = ks[0], kmax = ks[0];
    for (uint32_t i = 1; i < nk; i++) { if (ks[i] < kmin) kmin = ks[i]; if (ks[i] > kmax) kmax = ks[i]; }
    if (kmin < 1) { nm_set_error("k-mer lengths must be >= 1"); return NM_E_ARGUMENT; }
    if (range_mode && kmin == kmax) { nm_set_error("math domain error: a k-mer range needs two different lengths"); return NM_E_ARGUMENT; }
    std::vector<std::string> include, exclude;
    for (uint32_t i = 0; i < n_include; i++) include.emplace_back(include_ids[i]);
    for (uint32_t i = 0; i < n_exclude; i++) exclude.emplace_back(exclude_ids[i]);
    const char *stream_only = getenv("NEWMAP_AMD_STREAMING_DRIVER");
    if (!(stream_only && stream_only[0] == '1')) {
        std::vector<uint64_t> info;
        int rc = fast_run(ix, fasta_path, out_dir, ks, nk, range_mode, use_revcomp, batch, include, exclude, cb, user, total, rank, world, &info);
        if (rc == NM_OK && world == 1) {
            // a record that is one of the indexed records holds no absent k-mer; every other one is searched again by the
            // exact guard (nm_hash.h).  With several ranks the caller joins the ranks' shares first (newmap_amd/parallel.py).
            std::vector<uint8_t> flags;
            if (unverified_records(ix, info, flags)) rc = guard_pass(ix, fasta_path, ks, nk, range_mode, use_revcomp, batch, include, exclude, flags);
        }
        if (rec_info_out) *rec_info_out = info;
        if (rc != -1) return rc;
    }
    if (world > 1) {
        nm_set_error("the sharded native driver needs a FASTA file it can map or inflate: %s", fasta_path);
        return NM_E_ARGUMENT;
    }
    Driver d;
    d.ix = ix;
    d.device = (int)nm_index_info(ix, 10);
    d.ks.assign(ks, ks + nk);
    d.range_mode = range_mode != 0;
    d.use_rc = use_revcomp != 0;
    d.kmin = kmin;
    d.kmax = kmax;
    d.elem_bytes = d.kmax <= 0xFF ? 1 : (d.kmax <= 0xFFFF ? 2 : 4);      // newmap/search.py:204-212
    d.suffix = d.elem_bytes == 1 ? "uint8" : (d.elem_bytes == 2 ? "uint16" : "uint32");
    d.batch = batch >= 64 ? batch & ~63ull : batch;          // (segments start at multiples of 64 bases of their record: their fingerprints join, nm_hash.h)
    d.lookahead = d.kmax - 1;                                            // newmap/search.py:229
    d.out_dir = out_dir;
    d.include = include;
    d.exclude = exclude;
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
        {   // (streaming front-end: the same record check)
            std::vector<uint64_t> info;
            for (size_t o = 0; o < d.rec_hash.size(); o++) { info.push_back(d.rec_len[o]); info.push_back(d.rec_hash[o]); info.push_back(d.rec_searched[o]); }
            std::vector<uint8_t> flags;
            if (unverified_records(ix, info, flags)) rc = guard_pass(ix, fasta_path, ks, nk, range_mode, use_revcomp, batch, include, exclude, flags);
            if (rec_info_out) *rec_info_out = info;
        }
        if (!d.any_processed) {
            nm_set_error(d.include.empty() ? "The excluded sequences were too strict and nothing was processed"
                                           : "None of the included sequences were found");
            if (!d.include.empty() || !d.exclude.empty()) rc = NM_E_ARGUMENT;
        }
    }
    for (auto &s : d.slots) free_slot(s);
    (void)nm_stream_release(ix, d.stream);
    (void)hipStreamDestroy(d.stream);
    return rc;
}

extern "C" int nm_search_fasta(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks,
                               uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                               const char *const *include_ids, uint32_t n_include,
                               const char *const *exclude_ids, uint32_t n_exclude,
                               nm_record_callback cb, void *user, nm_search_summary *total) {
    return search_fasta_impl(ix, fasta_path, out_dir, ks, nk, range_mode, use_revcomp, batch, include_ids, n_include, exclude_ids, n_exclude,
                             cb, user, total, 0, 1);
}

extern "C" int nm_search_fasta_shard(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks,
                                     uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                                     const char *const *include_ids, uint32_t n_include,
                                     const char *const *exclude_ids, uint32_t n_exclude,
                                     nm_record_callback cb, void *user, nm_search_summary *total, int rank, int world) {
    return search_fasta_impl(ix, fasta_path, out_dir, ks, nk, range_mode, use_revcomp, batch, include_ids, n_include, exclude_ids, n_exclude,
                             cb, user, total, rank, world);
}

// the sharded search with the record fingerprints handed back: rec_info[3 i + 0 .. 2] = {length, fingerprint summed over THIS
// rank's units, searched} of record i (records with data, in file order), *n_records = their number (rec_info holds up to
// `capacity` records).  The caller adds the ranks' fingerprints (mod 2^64), asks nm_index_has_record and runs nm_guard_fasta
// over the records that are not indexed.
extern "C" int nm_search_fasta_shard_ex(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks,
                                        uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                                        const char *const *include_ids, uint32_t n_include,
                                        const char *const *exclude_ids, uint32_t n_exclude,
                                        nm_record_callback cb, void *user, nm_search_summary *total, int rank, int world,
                                        uint64_t *rec_info, uint64_t capacity, uint64_t *n_records) {
    std::vector<uint64_t> info;
    const int rc = search_fasta_impl(ix, fasta_path, out_dir, ks, nk, range_mode, use_revcomp, batch, include_ids, n_include, exclude_ids, n_exclude,
                                     cb, user, total, rank, world, &info);
    if (n_records) *n_records = info.size() / 3;
    if (rec_info) for (size_t i = 0; i < info.size() && i < 3 * capacity; i++) rec_info[i] = info[i];
    return rc;
}

// the exact guard over the records with flags[i] != 0 (records with data, in file order; nm_guard_segment_dev per segment)
extern "C" int nm_guard_fasta(nm_index *ix, const char *fasta_path, const uint32_t *ks, uint32_t nk, int range_mode, int use_revcomp,
                              uint64_t batch, const char *const *include_ids, uint32_t n_include, const char *const *exclude_ids,
                              uint32_t n_exclude, const uint8_t *flags, uint64_t n_flags) {
    if (!ix || !fasta_path || !ks || nk == 0 || (!flags && n_flags)) { nm_set_error("null argument"); return NM_E_ARGUMENT; }
    if (batch == 0) { nm_set_error("batch must be positive"); return NM_E_ARGUMENT; }
    std::vector<std::string> include, exclude;
    for (uint32_t i = 0; i < n_include; i++) include.emplace_back(include_ids[i]);
    for (uint32_t i = 0; i < n_exclude; i++) exclude.emplace_back(exclude_ids[i]);
    return guard_pass(ix, fasta_path, ks, nk, range_mode, use_revcomp, batch, include, exclude, std::vector<uint8_t>(flags, flags + n_flags));
}

// tests/hostsim/driver_sim.cpp -- TEST INFRASTRUCTURE ONLY: the engine entry points newmap_amd/csrc/nm_driver.hip calls, as
// host stubs, so that the driver's host logic runs under ThreadSanitizer (see fake_hip/hip/hip_runtime.h).  The "search"
// writes a function of the bytes that depends on the position's neighbourhood (so a segment cut in the wrong place, a
// missing lookahead or a write at the wrong file offset shows), the "fingerprint" is the real one (nm_hash.h).
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/newmap_amd.h"
#include "../../newmap_amd/csrc/nm_hash.h"

struct nm_index { std::vector<std::pair<uint64_t, uint64_t>> records; uint64_t guard_segments = 0; int in_call = 0; };

static thread_local char g_err[2048];
extern "C++" void nm_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
extern "C" const char *nm_last_error(void) { return g_err; }

static inline int code_of(uint8_t b) { const uint8_t u = b & 0xDF; return u == 'A' ? 0 : u == 'C' ? 1 : u == 'G' ? 2 : u == 'T' ? 3 : 4; }

static uint64_t fingerprint(const uint8_t *seq, uint64_t n) {
    uint64_t h = 0, pw = 1;
    for (uint64_t w = 0; w * 64 < n; w++) {
        uint64_t lo = 0, hi = 0, amb = 0;
        for (uint64_t j = 0; j < 64 && w * 64 + j < n; j++) {
            const int c = code_of(seq[w * 64 + j]);
            if (c > 3) amb |= 1ULL << j; else { lo |= (uint64_t)(c & 1) << j; hi |= (uint64_t)(c >> 1) << j; }
        }
        h += nm_hash_word(lo, hi, amb) * pw;
        pw *= NM_HASH_R;
    }
    return h;
}

extern "C" {
uint64_t nm_index_info(const nm_index *ix, int what) { return what == 23 ? ix->guard_segments : 0; }
int nm_index_has_record(const nm_index *ix, uint64_t length, uint64_t fp) {
    for (auto &r : ix->records) if (r.first == length && r.second == fp) return 1;
    return 0;
}
int nm_stream_release(nm_index *, void *) { return 0; }

// element of position p: depends on the bytes at p and p + kmax - 1 (inside the segment's lookahead) -- 0 if either is missing
static int fake_search(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers, uint32_t kmax, int elem_bytes, void *d_out, uint64_t *st) {
    if (++ix->in_call != 1) { nm_set_error("two engine calls at once on one handle"); return NM_E_DEVICE; }   // (the driver must serialise them)
    const uint8_t *s = (const uint8_t *)d_seq;
    memset(st, 0, NM_STATUS_WORDS * 8);
    st[2] = ~0ULL;
    for (uint64_t p = 0; p < num_kmers; p++) {
        uint32_t v = 0;
        if (code_of(s[p]) > 3) st[0]++;
        else if (p + kmax - 1 < seq_len) v = 1 + (uint32_t)((s[p] * 7u + s[p + kmax - 1] * 13u + (uint32_t)kmax) % 200u);
        if (elem_bytes == 1) ((uint8_t *)d_out)[p] = (uint8_t)v; else if (elem_bytes == 2) ((uint16_t *)d_out)[p] = (uint16_t)v; else ((uint32_t *)d_out)[p] = v;
    }
    st[NM_STATUS_HASH] = fingerprint(s, num_kmers);
    --ix->in_call;
    return NM_OK;
}
int nm_min_unique_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers, uint32_t, uint32_t kmax, int, int elem_bytes,
                              void *d_out, uint64_t *d_status, void *) { return fake_search(ix, d_seq, seq_len, num_kmers, kmax, elem_bytes, d_out, d_status); }
int nm_fixed_k_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks, uint32_t nk, int, int elem_bytes,
                           void *d_out, uint64_t *d_status, void *) {
    uint32_t kmax = 0;
    for (uint32_t i = 0; i < nk; i++) kmax = ks[i] > kmax ? ks[i] : kmax;
    return fake_search(ix, d_seq, seq_len, num_kmers, kmax, elem_bytes, d_out, d_status);
}
// the guard "finds" an absent k-mer at every byte 'X' (so tests can make it raise) and counts its segments
int nm_guard_segment_dev(nm_index *ix, const void *d_seq, uint64_t, uint64_t num_kmers, const uint32_t *, uint32_t, int, uint32_t, int, uint64_t *st, void *) {
    if (++ix->in_call != 1) { nm_set_error("two engine calls at once on one handle"); return NM_E_DEVICE; }
    memset(st, 0, NM_STATUS_WORDS * 8);
    st[2] = ~0ULL;
    ix->guard_segments++;
    const uint8_t *s = (const uint8_t *)d_seq;
    for (uint64_t p = 0; p < num_kmers; p++) if (s[p] == 'X') { st[1] = 1; st[2] = p; break; }
    --ix->in_call;
    return NM_OK;
}

// harness entry points (ctypes)
nm_index *ds_index_new(void) { return new nm_index(); }
void ds_index_add_record(nm_index *ix, const uint8_t *seq, uint64_t n) { ix->records.push_back({n, fingerprint(seq, n)}); }
void ds_index_free(nm_index *ix) { delete ix; }
}

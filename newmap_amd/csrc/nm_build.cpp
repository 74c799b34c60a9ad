// nm_build.cpp -- host-side index construction for the MI355X engine.
//
// Replaces generate_fm_index() of the reference (src/newmap-generate-index.c:11-58,82-104 ->
// awFmCreateIndexFromFasta): FASTA -> both-strand run text (nm_format.h) -> suffix array
// (SA-IS, nm_sais.hpp) -> BWT -> 32-byte rank blocks + strand blocks + separator list -> file.
// The reference indexes the forward strand only and searches the reverse complement as a
// second query (newmap/search.py:677-697); indexing both strands turns that into one search
// and lets the device extend a k-mer one base at a time.
//
// FASTA reading follows newmap/fasta.py:47,59,75 (the `search` side of the reference) so that
// `index` and `search` agree on what a record is: lines are right-stripped, '>' or ';' opens a
// record, everything else is sequence data; records without data do not exist.
#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <zlib.h>

#include "../../include/newmap_amd.h"
#include "nm_format.h"
#include "nm_hash.h"
#include "nm_internal.h"
#include "nm_sais.hpp"
#include "nm_pdsa.hpp"

static thread_local char g_err[1024] = "";

void nm_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *nm_last_error(void) { return g_err; }
extern "C" const char *nm_version(void) { return "newmap_amd 0.2 (gfx950, index format 2)"; }

namespace {

// text symbols handed to the suffix sorter
enum : uint8_t { SYM_END = 0, SYM_SEP = 1, SYM_A = 2 };   // A,C,G,T = 2..5

struct Lut {
    uint8_t code[256];
    Lut() {
        memset(code, 0xFF, sizeof code);
        code[(int)'A'] = code[(int)'a'] = 0;
        code[(int)'C'] = code[(int)'c'] = 1;
        code[(int)'G'] = code[(int)'g'] = 2;
        code[(int)'T'] = code[(int)'t'] = 3;
    }
};
const Lut g_lut;

inline bool is_space(unsigned char c) {      // what bytes.rstrip() removes
    return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f';
}

struct FastaText {
    std::vector<uint8_t> f;        // forward run text F: runs of SYM_A.. each followed by SYM_SEP
    std::vector<nm_record_entry> records;   // (length, fingerprint) of every record with data (nm_hash.h)
    uint64_t rec_len = 0, rec_hash = 0, rec_pw = 1, rec_lo = 0, rec_hi = 0, rec_amb = 0;
    uint64_t n_records = 0, raw_bases = 0, n_runs = 0;
    uint64_t base_count[4] = {0, 0, 0, 0};
    bool in_run = false;
    bool record_has_data = false;

    void end_run() {
        if (in_run) { f.push_back(SYM_SEP); n_runs++; in_run = false; }
    }
    void data(const unsigned char *p, size_t len) {
        if (!len) return;
        if (!record_has_data) { record_has_data = true; n_records++; }
        raw_bases += len;
        // a line yields at most one symbol per byte: write through a raw pointer, trim afterwards
        const size_t old = f.size();
        f.resize(old + len);
        uint8_t *out = f.data() + old;
        uint64_t cnt[4] = {0, 0, 0, 0};
        bool run = in_run;
        uint64_t runs = 0;
        uint64_t wlo = rec_lo, whi = rec_hi, wamb = rec_amb;      // the record's current 64-base word (nm_hash.h)
        unsigned bit = (unsigned)(rec_len & 63u);
        for (size_t i = 0; i < len; i++) {
            const uint8_t c = g_lut.code[p[i]];
            if (c != 0xFF) { *out++ = (uint8_t)(SYM_A + c); cnt[c]++; run = true; wlo |= (uint64_t)(c & 1u) << bit; whi |= (uint64_t)(c >> 1) << bit; }
            else { wamb |= 1ULL << bit; if (run) { *out++ = SYM_SEP; runs++; run = false; } }
            if (++bit == 64) { rec_hash += nm_hash_word(wlo, whi, wamb) * rec_pw; rec_pw *= NM_HASH_R; wlo = whi = wamb = 0; bit = 0; }
        }
        rec_lo = wlo; rec_hi = whi; rec_amb = wamb;
        rec_len += len;
        f.resize((size_t)(out - f.data()));
        in_run = run;
        n_runs += runs;
        for (int c = 0; c < 4; c++) base_count[c] += cnt[c];
    }
    void close_record() {
        if (record_has_data) {
            if (rec_len & 63u) rec_hash += nm_hash_word(rec_lo, rec_hi, rec_amb) * rec_pw;   // the last, partial word
            records.push_back(nm_record_entry{rec_len, rec_hash});
        }
        rec_len = 0; rec_hash = 0; rec_pw = 1; rec_lo = rec_hi = rec_amb = 0;
    }
    void header() { end_run(); close_record(); record_has_data = false; }
    void line(const unsigned char *p, size_t len) {
        while (len && is_space(p[len - 1])) len--;
        if (len && (p[0] == '>' || p[0] == ';')) header();
        else data(p, len);
    }
    void finish() { end_run(); close_record(); record_has_data = false; }
};

int read_fasta(const char *path, FastaText &ft) {
    gzFile gz = gzopen(path, "rb");          // transparent for plain files
    if (!gz) { nm_set_error("Could not open fasta file to create index: %s", path); return NM_E_FILE_OPEN; }
    gzbuffer(gz, 1 << 20);
    {   // room for the whole text at once (a plain FASTA is a little larger than its bases)
        struct stat sb;
        if (stat(path, &sb) == 0 && sb.st_size > 0 && gzdirect(gz)) ft.f.reserve((size_t)sb.st_size);
    }
    std::vector<unsigned char> buf(1 << 22);
    std::vector<unsigned char> carry;
    for (;;) {
        int got = gzread(gz, buf.data(), (unsigned)buf.size());
        if (got < 0) { gzclose(gz); nm_set_error("read error in %s", path); return NM_E_FILE_OPEN; }
        if (got == 0) break;
        size_t start = 0;
        for (;;) {
            const unsigned char *nl = (const unsigned char *)memchr(buf.data() + start, '\n', (size_t)got - start);
            if (!nl) break;
            const size_t i = (size_t)(nl - buf.data());
            if (!carry.empty()) {
                carry.insert(carry.end(), buf.begin() + start, buf.begin() + i);
                ft.line(carry.data(), carry.size());
                carry.clear();
            } else {
                ft.line(buf.data() + start, i - start);
            }
            start = i + 1;
        }
        carry.insert(carry.end(), buf.begin() + start, buf.begin() + got);
    }
    if (!carry.empty()) ft.line(carry.data(), carry.size());
    ft.finish();
    gzclose(gz);
    return NM_OK;
}

thread_local uint8_t *g_lcp_buffer = nullptr;
thread_local bool g_lcp_done = false;

// bases the suffixes at p and q share (a separator matches nothing), capped
static inline uint8_t lcp_capped(const uint8_t *T, uint64_t p, uint64_t q) {
    uint32_t h = 0;
    while (h < NM_LCP_CAP) {
        const uint8_t a = T[p + h];
        if (a < SYM_A || a != T[q + h]) break;               // (T ends with SYM_END: the loop stops inside the text)
        h++;
    }
    return (uint8_t)h;
}

template <class I>
int build_and_write(const FastaText &ft, const char *index_path, uint8_t sa_ratio, uint8_t seed_len,
                    nm_sa32_provider provider = nullptr, void *provider_ctx = nullptr, nm_bwt_provider bwt_provider = nullptr) {
    const uint64_t nf = ft.f.size();
    const uint64_t n = 2 * nf + 1;
    // T = F . RC . '#'
    nm::PdBuf<uint8_t> T(n);                                   // (uninitialised: every byte is written below)
    if (nf) {
        // F, then F reversed without its trailing separator with the bases complemented, closed by a separator
#pragma omp parallel for schedule(static) num_threads(nm::pd_threads())
        for (int64_t i = 0; i < (int64_t)nf; i++) {
            T[(uint64_t)i] = ft.f[(uint64_t)i];
            if ((uint64_t)i + 1 < nf) {
                const uint8_t c = ft.f[nf - 2 - (uint64_t)i];
                T[nf + (uint64_t)i] = c >= SYM_A ? (uint8_t)(SYM_A + 3 - (c - SYM_A)) : c;
            }
        }
        T[2 * nf - 1] = SYM_SEP;
    }
    T[n - 1] = SYM_END;

    // suffix array: parallel prefix doubling when there are cores to use, SA-IS otherwise
    // (NEWMAP_AMD_SA=sais|pd forces one; both give the same array, tests/test_index_and_core_cpu.py)
    const bool verbose = nm::pd_verbose();
    double tv = nm::pd_now();
    const int nt = nm::pd_threads();
    nm::PdBuf<uint8_t> bw(n);                                  // BWT symbol (low bits) | "suffix starts in the RC half" (bit 7)
    // LCP bytes (nm_format.h): NEWMAP_AMD_LCP=0 builds an index without them
    const char *lcp_env = getenv("NEWMAP_AMD_LCP");
    const bool want_lcp = !(lcp_env && lcp_env[0] == '0');
    nm::PdBuf<uint8_t> lcp(want_lcp ? n + 1 : 1);
    g_lcp_buffer = want_lcp ? lcp.data() : nullptr;
    g_lcp_done = false;
    struct LcpReset { ~LcpReset() { g_lcp_buffer = nullptr; } } lcp_reset;
    bool have_bw = false;
    if (bwt_provider) {
        // suffix sort AND the gather of the BWT on the device (nm_build_device.hip): only n bytes come back
        const int rc = bwt_provider(T.data(), n, nf, bw.data(), provider_ctx);
        if (rc == NM_OK) {
            have_bw = true;
            if (verbose) { fprintf(stderr, "[build] suffix array + BWT on the device: %.2fs\n", nm::pd_now() - tv); tv = nm::pd_now(); }
        } else if (rc != NM_E_ALLOC && rc != NM_E_TOO_LARGE) {
            return rc;
        } else if (verbose) {
            fprintf(stderr, "[build] device suffix sort declined (%s): host sorter\n", nm_last_error());
        }
    }
    nm::PdBuf<I> SA(have_bw ? 1 : n);
    if (!have_bw) {
        const char *want = getenv("NEWMAP_AMD_SA");
        const bool use_pd = want ? strcmp(want, "pd") == 0 : (nm::pd_threads() > 1 && n > (1u << 16));
        if (provider && sizeof(I) == 4) {
            // suffix array computed elsewhere (the device builder, nm_build_device.hip)
            int rc = provider(T.data(), n, (int32_t *)SA.data(), provider_ctx);
            if (rc != NM_OK) return rc;
        } else if (use_pd) nm::pd_suffix_array<I>(T.data(), n, SA.data());
        else nm::sais<uint8_t, I>(T.data(), SA.data(), (I)n, (I)6);
        if (verbose) { fprintf(stderr, "[build] suffix array (%s, %d threads): %.2fs\n", provider ? "device prefix doubling" : (use_pd ? "prefix doubling" : "SA-IS"), nm::pd_threads(), nm::pd_now() - tv); tv = nm::pd_now(); }
        if (want_lcp && !g_lcp_done) {                         // (a device provider has filled them from its own copy of the array)
            lcp[0] = 0;
#pragma omp parallel for schedule(dynamic, 1 << 16) num_threads(nm::pd_threads())
            for (int64_t j = 1; j < (int64_t)n; j++) lcp[(uint64_t)j] = lcp_capped(T.data(), (uint64_t)SA[(uint64_t)j], (uint64_t)SA[(uint64_t)j - 1]);
            g_lcp_done = true;
            if (verbose) { fprintf(stderr, "[build] LCP bytes on the host: %.2fs\n", nm::pd_now() - tv); tv = nm::pd_now(); }
        }
    }
    const bool have_lcp = want_lcp && g_lcp_done;
    if (have_lcp) { lcp[0] = 0; lcp[n] = 0; }

    nm_file_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, NM_MAGIC, 8);
    h.version = NM_FORMAT_VERSION;
    h.header_bytes = sizeof h;
    h.n = n;
    h.n_fwd = nf;
    h.n_sep = 2 * ft.n_runs + 1;
    for (int c = 0; c < 4; c++) h.base_count[c] = ft.base_count[c] + ft.base_count[3 - c];
    h.n_rank_blocks = n / 64 + 1;
    h.n_strand_blocks = n / 64 + 1;
    h.n_records = ft.n_records;
    h.raw_bases = ft.raw_bases;
    h.n_runs = ft.n_runs;
    h.sa_ratio = sa_ratio;
    h.seed_len = seed_len;
    h.n_super = (n >> NM_SUPER_SHIFT) + 1;
    if (h.n_super > NM_MAX_SUPER) { nm_set_error("text of %llu symbols is too large", (unsigned long long)n); return NM_E_TOO_LARGE; }

    // ---- BWT symbol (low bits) and "suffix starts in the RC half" (bit 7) per suffix-array position
    const uint64_t nblk = h.n_rank_blocks;
    std::vector<uint64_t> part((size_t)(nt + 1) * 5, 0);          // per block range: A,C,G,T,rc-half
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const uint64_t i0 = std::min<uint64_t>(nblk * (uint64_t)t / nt * 64, n);
        const uint64_t i1 = std::min<uint64_t>(nblk * (uint64_t)(t + 1) / nt * 64, n);
        uint64_t cnt[5] = {0, 0, 0, 0, 0};
        for (uint64_t i = i0; i < i1; i++) {
            if (!have_bw) {
                const uint64_t p = (uint64_t)SA[i];
                const uint8_t c0 = T[p ? p - 1 : n - 1];
                bw[i] = (uint8_t)(c0 | ((p >= nf && p < 2 * nf) ? 0x80 : 0));
            }
            const uint8_t ch = bw[i] & 0x7F;
            if (ch >= SYM_A) cnt[ch - SYM_A]++;
            cnt[4] += (bw[i] >> 7);
        }
        for (int c = 0; c < 5; c++) part[(size_t)(t + 1) * 5 + c] = cnt[c];
    }
    SA.release();
    for (int t = 0; t < nt; t++)
        for (int c = 0; c < 5; c++) part[(size_t)(t + 1) * 5 + c] += part[(size_t)t * 5 + c];

    // ---- blocks: absolute counts first, made superblock-relative afterwards
    std::vector<nm_rank_block> rank(nblk);
    std::vector<nm_strand_block> strand(h.n_strand_blocks);
    nm::PdBuf<uint64_t> absv(nblk * 4);
    std::vector<std::vector<uint64_t>> tsep((size_t)nt);
#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const uint64_t b0 = nblk * (uint64_t)t / nt, b1 = nblk * (uint64_t)(t + 1) / nt;
        uint64_t run[4], rc_before = part[(size_t)t * 5 + 4];
        for (int c = 0; c < 4; c++) run[c] = part[(size_t)t * 5 + c];
        for (uint64_t b = b0; b < b1; b++) {
            for (int c = 0; c < 4; c++) absv[(size_t)b * 4 + c] = run[c];
            uint64_t lo = 0, hi = 0, bits = 0;
            uint32_t flag = 0;
            const uint64_t base = b * 64, e = std::min<uint64_t>(base + 64, n);
            for (uint64_t i = base; i < e; i++) {
                const uint8_t v = bw[i], ch = v & 0x7F;
                const uint64_t bit = 1ULL << (i & 63);
                if (ch >= SYM_A) {
                    const unsigned c = ch - SYM_A;
                    run[c]++;
                    if (c & 1) lo |= bit;
                    if (c & 2) hi |= bit;
                } else {
                    tsep[t].push_back(i);
                    flag = NM_SEP_FLAG;
                }
                if (v & 0x80) bits |= bit;
            }
            rank[b].cnt[0] = flag; rank[b].cnt[1] = rank[b].cnt[2] = rank[b].cnt[3] = 0;
            rank[b].lo = lo; rank[b].hi = hi;
            strand[b].before = rc_before;
            strand[b].bits = bits;
            rc_before += (uint64_t)__builtin_popcountll(bits);
        }
    }
    const uint64_t blocks_per_super = 1ULL << (NM_SUPER_SHIFT - 6);
    for (uint64_t sb = 0; sb < h.n_super; sb++)
        for (int c = 0; c < 4; c++) h.super_cnt[sb][c] = absv[(size_t)(sb * blocks_per_super) * 4 + c];
#pragma omp parallel for schedule(static) num_threads(nt)
    for (int64_t b = 0; b < (int64_t)nblk; b++) {
        const uint64_t sb = (uint64_t)b / blocks_per_super;
        for (int c = 0; c < 4; c++)
            rank[b].cnt[c] |= (uint32_t)(absv[(size_t)b * 4 + c] - h.super_cnt[sb][c]);
    }
    if (verbose) { fprintf(stderr, "[build] BWT + blocks: %.2fs\n", nm::pd_now() - tv); tv = nm::pd_now(); }
    std::vector<uint64_t> sep;
    sep.reserve(h.n_sep);
    for (auto &v : tsep) sep.insert(sep.end(), v.begin(), v.end());
    if (sep.size() != h.n_sep) { nm_set_error("internal error: separator count mismatch"); return NM_E_FILE_WRITE; }

    h.off_rank = sizeof h;
    h.off_strand = h.off_rank + rank.size() * sizeof(nm_rank_block);
    h.off_sep = h.off_strand + strand.size() * sizeof(nm_strand_block);
    h.off_records = h.off_sep + sep.size() * sizeof(uint64_t);
    if (ft.records.size() != h.n_records) { nm_set_error("internal error: record count mismatch"); return NM_E_FILE_WRITE; }
    h.file_bytes = h.off_records + ft.records.size() * sizeof(nm_record_entry);
    if (have_lcp) { h.off_lcp = h.file_bytes; h.file_bytes += n + 1; }

    FILE *fp = fopen(index_path, "wb");        // overwrite, like the reference at its pinned version
    if (!fp) { nm_set_error("Could not write index file %s: %s", index_path, strerror(errno)); return NM_E_FILE_WRITE; }
    bool ok = fwrite(&h, sizeof h, 1, fp) == 1 &&
              fwrite(rank.data(), sizeof(nm_rank_block), rank.size(), fp) == rank.size() &&
              fwrite(strand.data(), sizeof(nm_strand_block), strand.size(), fp) == strand.size() &&
              (sep.empty() || fwrite(sep.data(), sizeof(uint64_t), sep.size(), fp) == sep.size()) &&
              (ft.records.empty() || fwrite(ft.records.data(), sizeof(nm_record_entry), ft.records.size(), fp) == ft.records.size()) &&
              (!have_lcp || fwrite(lcp.data(), 1, n + 1, fp) == n + 1);
    ok = (fclose(fp) == 0) && ok;
    if (!ok) { nm_set_error("Could not write index file %s", index_path); return NM_E_FILE_WRITE; }
    return NM_OK;
}

}  // namespace

int nm_index_build_impl(const char *fasta_path, const char *index_path, uint8_t sa_ratio, uint8_t seed_len,
                        nm_sa32_provider provider, void *provider_ctx, nm_bwt_provider big_provider) {
    if (!fasta_path || !index_path) { nm_set_error("null path"); return NM_E_ARGUMENT; }
    if (seed_len > 16) { nm_set_error("seed length %d is larger than the supported maximum 16", (int)seed_len); return NM_E_ARGUMENT; }
    try {
        {   // gzopen succeeds lazily on some platforms; probe with fopen for the reference's error
            FILE *probe = fopen(fasta_path, "rb");
            if (!probe) { nm_set_error("Could not open fasta file to create index: %s", fasta_path); return NM_E_FILE_OPEN; }
            fclose(probe);
        }
        FastaText ft;
        int rc = read_fasta(fasta_path, ft);
        if (rc != NM_OK) return rc;
        const uint64_t n = 2 * (uint64_t)ft.f.size() + 1;
        // NEWMAP_AMD_DEVICE_SA=large (tests): the large-text device path on a small text
        const char *force = getenv("NEWMAP_AMD_DEVICE_SA");
        const bool force_large = big_provider && force && strcmp(force, "large") == 0;
        if (n < (1ULL << 31) - 8 && !force_large) return build_and_write<int32_t>(ft, index_path, sa_ratio, seed_len, provider, provider_ctx);
        if (n < (1ULL << 31) - 8) return build_and_write<int32_t>(ft, index_path, sa_ratio, seed_len, nullptr, provider_ctx, big_provider);
        return build_and_write<int64_t>(ft, index_path, sa_ratio, seed_len, nullptr, provider_ctx, big_provider);   // host sorter unless the device takes it
    } catch (const std::bad_alloc &) {
        nm_set_error("Could not allocate enough memory to create index");
        return NM_E_ALLOC;
    } catch (const std::exception &e) {
        nm_set_error("index build failed: %s", e.what());
        return NM_E_FILE_WRITE;
    }
}

uint8_t *nm_build_lcp_buffer(void) { return g_lcp_buffer; }
void nm_build_lcp_done(void) { g_lcp_done = true; }

extern "C" int nm_index_build(const char *fasta_path, const char *index_path, uint8_t sa_ratio, uint8_t seed_len) {
    return nm_index_build_impl(fasta_path, index_path, sa_ratio, seed_len, nullptr, nullptr);
}

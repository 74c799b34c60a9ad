// nm_tables.hip.h -- tables built at nm_index_open (seed levels, quad tables, LF blocks), open / close / info / options (part
// of nm_engine.hip).
// build the 4^s table on the device; launches are sliced so that grid * block stays below 2^32
static int nm_seed_launch(nm_index *ix, const nm_view &v, const uint64_t *parent, uint64_t *table, uint32_t s) {
    const uint64_t n_slots = 1ULL << (2 * s), slice = 1ULL << 30;
    for (uint64_t first = 0; first < n_slots; first += slice) {
        const uint64_t m = n_slots - first < slice ? n_slots - first : slice;
        const dim3 grid(nm_grid(m)), block(NM_BLOCK);
        if (parent) {
            if (ix->big) hipLaunchKernelGGL(k_seed_level<true>, grid, block, 0, ix->stream, v, parent, table, first, n_slots, s);
            else         hipLaunchKernelGGL(k_seed_level<false>, grid, block, 0, ix->stream, v, parent, table, first, n_slots, s);
        } else {
            if (ix->big) hipLaunchKernelGGL(k_seed<true>, grid, block, 0, ix->stream, v, table, first, n_slots, s);
            else         hipLaunchKernelGGL(k_seed<false>, grid, block, 0, ix->stream, v, table, first, n_slots, s);
        }
        HIP_TRY(hipGetLastError());
    }
    return NM_OK;
}

// build the 4^s table on the device: level 8 entry by entry, every further level from the one
// below it (launches sliced so that grid * block stays below 2^32)
// NEWMAP_AMD_VERBOSE=1: phase timings of nm_index_open on stderr
static bool nm_verbose() { const char *v = getenv("NEWMAP_AMD_VERBOSE"); return v && *v && *v != '0'; }
static double nm_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define NM_PHASE(t0, what) do { if (nm_verbose()) { fprintf(stderr, "[open] %s: %.3fs\n", what, nm_now() - (t0)); (t0) = nm_now(); } } while (0)

// quad table for cores of m bases, from the seed table of that length (a level of the seed-table build):
// 4^m entries x 128 bytes
static int nm_build_quad(nm_index *ix, const uint64_t *level_table, uint32_t m, bool small = false) {
    if (!small) { ix->view.quad = nullptr; ix->view.quad_m = 0; }
    else ix->quad_small_m = 0;
    if (!level_table || m < 3 || m > 16 || ix->h.n < 2) return NM_OK;
    const uint64_t n_cores = 1ULL << (2 * m);
    double tq = nm_now();
    void **slot = small ? &ix->d_quad_small : &ix->d_quad;
    if (hipMalloc(slot, n_cores * NM_QUAD_WORDS * 8) != hipSuccess) {     // (someone else holds the memory: go on without the table)
        (void)hipGetLastError();
        *slot = nullptr;
        if (nm_verbose()) fprintf(stderr, "[open] quad table of %llu GB does not fit: range mode runs on the seed table\n",
                                  (unsigned long long)(n_cores * NM_QUAD_WORDS * 8 >> 30));
        return NM_OK;
    }
    NM_PHASE(tq, "quad table hipMalloc");
    ix->device_bytes += n_cores * NM_QUAD_WORDS * 8;
    nm_view v = ix->view;
    v.seed = level_table;
    v.seed_len = m;
    const uint64_t slice = 1ULL << 30;
    for (uint64_t first = 0; first < n_cores; first += slice) {
        const uint64_t cnt = n_cores - first < slice ? n_cores - first : slice;
        if (ix->big) hipLaunchKernelGGL(k_quad_build<true>, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, v, (uint64_t *)*slot, first, n_cores, m);
        else         hipLaunchKernelGGL(k_quad_build<false>, dim3(nm_grid(cnt)), dim3(NM_BLOCK), 0, ix->stream, v, (uint64_t *)*slot, first, n_cores, m);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(ix->stream));
    NM_PHASE(tq, "quad table kernels");
    if (small) ix->quad_small_m = m;
    else { ix->view.quad = (const uint64_t *)ix->d_quad; ix->view.quad_m = m; }
    return NM_OK;
}

// quad_m: also derive the quad table from the level of that length (0 = none)
static int nm_build_seed_table(nm_index *ix, uint32_t s, void **d_table, uint32_t quad_m = 0, uint32_t quad_small_m = 0) {
    const uint64_t n_slots = 1ULL << (2 * s);
    double ts = nm_now();
    HIP_TRY(hipMalloc(d_table, n_slots * sizeof(uint64_t)));
    NM_PHASE(ts, "seed table hipMalloc");
    ix->device_bytes += n_slots * sizeof(uint64_t);
    nm_view v = ix->view;
    v.seed = nullptr;
    v.seed_len = 0;
    const uint32_t s0 = s < 8 ? s : 8;
    void *cur = nullptr;
    int rc = NM_OK;
    for (uint32_t level = s0; level <= s && rc == NM_OK; level++) {
        void *dst = *d_table;
        if (level < s && hipMalloc(&dst, (8ULL << (2 * level))) != hipSuccess) { nm_set_error("hipMalloc failed for a seed level"); rc = NM_E_ALLOC; break; }
        rc = nm_seed_launch(ix, v, level == s0 ? nullptr : (const uint64_t *)cur, (uint64_t *)dst, level);
        if (rc == NM_OK && hipStreamSynchronize(ix->stream) != hipSuccess) { nm_set_error("seed table kernel failed"); rc = NM_E_DEVICE; }
        if (rc == NM_OK && level == quad_m) rc = nm_build_quad(ix, (const uint64_t *)dst, level);
        if (rc == NM_OK && level == quad_small_m && quad_small_m != quad_m) rc = nm_build_quad(ix, (const uint64_t *)dst, level, true);
        if (cur) (void)hipFree(cur);
        cur = level < s ? dst : nullptr;
    }
    if (cur) (void)hipFree(cur);
    NM_PHASE(ts, "seed table levels (incl. the quad table)");
    return rc;
}

// core length of the quad table: as long as the seed, at most 60 % of the HBM still free once the seed table is
// in place (4^m x 128 bytes: 137 GB for m = 15)
// Free HBM as the table sizing sees it: what the device reports, capped by what is left of the caller's budget for THIS index
// (nm_index_open_budget: several indexes of one search co-reside -- newmap/search.py:656-697 sums the counts of every index
// file -- each sized for its share).  0 = no budget.
static thread_local uint64_t g_open_budget = 0;
static hipError_t nm_free_hbm(const nm_index *ix, size_t *free_b, size_t *total_b) {
    const hipError_t e = hipMemGetInfo(free_b, total_b);
    if (e == hipSuccess && g_open_budget) {
        const uint64_t left = g_open_budget > ix->device_bytes ? g_open_budget - ix->device_bytes : 0;
        if (*free_b > left) *free_b = (size_t)left;
    }
    return e;
}

static uint32_t nm_auto_quad_len(const nm_index *ix, uint32_t s) {
    uint32_t m = s;
    size_t free_b = 0, total_b = 0;
    if (nm_free_hbm(ix, &free_b, &total_b) != hipSuccess) return 0;
    free_b -= free_b < (8ULL << (2 * s)) ? free_b : (8ULL << (2 * s));      // the seed table comes first
    while (m >= 8 && (128ULL << (2 * m)) > free_b / 5 * 3) m--;
    return m >= 8 ? m : 0;
}

// seed length that makes nearly all positions resolve in the table: two more bases than log4(n)
static uint32_t nm_auto_seed_len(const nm_index *ix) {
    uint32_t s = 1;
    while (s < 16 && (1ULL << (2 * s)) < ix->h.n) s++;     // s = ceil(log4 n)
    uint32_t bonus = 2;                                   // NEWMAP_AMD_SEED_BONUS: measurement knob
    if (const char *b = getenv("NEWMAP_AMD_SEED_BONUS")) bonus = (uint32_t)atoi(b);
    s = s + bonus > 16 ? 16 : s + bonus;
    if (s < 4) s = 4;
    size_t free_b = 0, total_b = 0;
    if (nm_free_hbm(ix, &free_b, &total_b) == hipSuccess)
        while (s > 4 && (8ULL << (2 * s)) > free_b / 4) s--;   // never more than a quarter of free HBM
    return s;
}

#include "nm_scan.hip.h"

// LF blocks: 64 B per 64 BWT rows (one 16-byte entry per base)
static int nm_build_lf_blocks(nm_index *ix) {
    const uint64_t n_blocks = ix->h.n / 64 + 1;
    HIP_TRY(hipMalloc(&ix->d_lfb, n_blocks * 4 * sizeof(nm_lf_entry)));
    ix->device_bytes += n_blocks * 4 * sizeof(nm_lf_entry);
    nm_view v = ix->view;
    v.lfb = nullptr;
    if (ix->big) hipLaunchKernelGGL(k_lf_blocks<true>, dim3(nm_grid(n_blocks)), dim3(NM_BLOCK), 0, ix->stream, v, (nm_lf_entry *)ix->d_lfb, n_blocks);
    else         hipLaunchKernelGGL(k_lf_blocks<false>, dim3(nm_grid(n_blocks)), dim3(NM_BLOCK), 0, ix->stream, v, (nm_lf_entry *)ix->d_lfb, n_blocks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->view.lfb = (const nm_lf_entry *)ix->d_lfb;
    return NM_OK;
}

static int nm_build_seed(nm_index *ix, uint32_t s, uint32_t quad_m = 0, uint32_t quad_small_m = 0) {
    ix->view.seed = nullptr;
    ix->view.seed_len = 0;
    if (s == 0 || ix->h.n < 2) return NM_OK;
    int rc = nm_build_seed_table(ix, s, &ix->d_seed, quad_m, quad_small_m);
    if (rc != NM_OK) return rc;
    ix->view.seed = (const uint64_t *)ix->d_seed;
    ix->view.seed_len = s;
    return NM_OK;
}

// Two-base LF blocks (nm_core.h): 4 bytes per BWT row; built last, where the memory left allows (NEWMAP_AMD_LF2=0: none)
static int nm_build_lf2(nm_index *ix) {
    ix->view.lf2 = nullptr;
    if (const char *e = getenv("NEWMAP_AMD_LF2")) if (e[0] == '0') return NM_OK;
    if (!ix->view.lfb || ix->h.n < 2) return NM_OK;
    const uint64_t n_blocks = ix->h.n / 64 + 1, n_chunks = (n_blocks + NM_LF2_CHUNK - 1) / NM_LF2_CHUNK;
    const uint64_t bytes = n_blocks * 16 * sizeof(nm_lf_entry);
    size_t free_b = 0, total_b = 0;
    if (nm_free_hbm(ix, &free_b, &total_b) != hipSuccess || bytes + (16ull << 30) > free_b) return NM_OK;
    double t0 = nm_now();
    void *table = nullptr;
    uint64_t *d_sums = nullptr;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc((void **)&d_sums, n_chunks * 16 * 8) != hipSuccess) {
        (void)hipGetLastError();
        if (table) (void)hipFree(table);
        return NM_OK;
    }
    ix->d_lf2 = table;                                             // (nm_index_close frees it, whatever happens below)
    ix->device_bytes += bytes;
    struct Sums { uint64_t *p; ~Sums() { if (p) (void)hipFree(p); } } sums_guard{d_sums};
    const nm_view v = ix->view;
    const uint64_t slice_blocks = 1ULL << 24;                      // (grid * block below 2^32: 64 lanes per block of rows)
    for (uint64_t first = 0; first < n_blocks; first += slice_blocks) {
        const uint64_t m = n_blocks - first < slice_blocks ? n_blocks - first : slice_blocks;
        if (ix->big) hipLaunchKernelGGL(k_lf2_bits<true>, dim3(nm_grid(m * 64)), dim3(NM_BLOCK), 0, ix->stream, v, (nm_lf_entry *)table, first, first + m);
        else         hipLaunchKernelGGL(k_lf2_bits<false>, dim3(nm_grid(m * 64)), dim3(NM_BLOCK), 0, ix->stream, v, (nm_lf_entry *)table, first, first + m);
        HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(k_lf2_chunk_sums, dim3(nm_grid(n_chunks * 16)), dim3(NM_BLOCK), 0, ix->stream, (const nm_lf_entry *)table, n_blocks, n_chunks, d_sums);
    std::vector<uint64_t> sums(n_chunks * 16);
    HIP_TRY(hipMemcpyAsync(sums.data(), d_sums, sums.size() * 8, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(hipStreamSynchronize(ix->stream));
    uint64_t run[16] = {0};
    for (uint64_t c = 0; c < n_chunks; c++)
        for (int d = 0; d < 16; d++) { const uint64_t x = sums[c * 16 + d]; sums[c * 16 + d] = run[d]; run[d] += x; }
    HIP_TRY(hipMemcpyAsync(d_sums, sums.data(), sums.size() * 8, hipMemcpyHostToDevice, ix->stream));
    if (ix->big) hipLaunchKernelGGL(k_lf2_finish<true>, dim3(nm_grid(n_chunks * 16)), dim3(NM_BLOCK), 0, ix->stream, v, (nm_lf_entry *)table, n_blocks, n_chunks, (const uint64_t *)d_sums);
    else         hipLaunchKernelGGL(k_lf2_finish<false>, dim3(nm_grid(n_chunks * 16)), dim3(NM_BLOCK), 0, ix->stream, v, (nm_lf_entry *)table, n_blocks, n_chunks, (const uint64_t *)d_sums);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ix->stream));
    ix->view.lf2 = (const nm_lf_entry *)table;
    if (nm_verbose()) fprintf(stderr, "[open] two-base LF blocks (%.1f GB): %.3fs\n", bytes / 1e9, nm_now() - t0);
    return NM_OK;
}

// The repeat dictionary (nm_core.h): every x-mer that occurs at least twice, x = ceil(log4 n) + 3 (<= 24), derived from the
// seed table of length s < x level by level (k_dict_expand), then hashed into buckets of 8 (k_dict_insert).  A genome
// whose repeated x-mers do not fit the memory at hand simply goes without (the second quad table and the seed walks
// remain).  NEWMAP_AMD_DICT=0: none; NEWMAP_AMD_DICT_LEN: another x.
static int nm_build_dict(nm_index *ix) {
    ix->view.dict = nullptr;
    ix->view.dict_len = ix->view.dict_bits = 0;
    const uint32_t s = ix->view.seed_len;
    if (const char *e = getenv("NEWMAP_AMD_DICT")) if (e[0] == '0') return NM_OK;
    if (!ix->view.seed || s < 8 || ix->h.n < 2) return NM_OK;
    uint32_t x = 1;
    while (x < 32 && (1ULL << (2 * x)) < ix->h.n) x++;
    x += 3;
    // (its strings must be longer than the second quad table's windows: it takes what the second chance leaves, sec. 4)
    if (ix->view.quad && ix->d_quad_small && x <= ix->view.quad_m + NM_QUAD_EXT) x = ix->view.quad_m + NM_QUAD_EXT + 1;    // (view.quad: the long cores, the second chance of launches that read the short ones)
    if (const char *e = getenv("NEWMAP_AMD_DICT_LEN")) x = (uint32_t)atoi(e);
    if (x > NM_DICT_MAX_LEN) x = NM_DICT_MAX_LEN;
    if (x <= s) return NM_OK;
    double td = nm_now();
    size_t free_b = 0, total_b = 0;
    if (nm_free_hbm(ix, &free_b, &total_b) != hipSuccess) return NM_OK;
    // room for the lists: a repeated L-mer has two occurrences at least (n / 2 strings), and a text without much repetition
    // has about 4^L (1 - e^-l (1 + l)), l = n / 4^L, of them at the first level -- twice that is allocated, the rest of the
    // memory stays untouched (a genome with more repeated strings than that goes without a dictionary)
    uint64_t cap = ix->h.n / 2 + 1024;
    {
        const double slots = pow(4.0, (double)(s + 1)), l = (double)ix->h.n / slots;
        const double est = slots * (1.0 - exp(-l) * (1.0 + l));
        if ((double)cap > 2.0 * est + 1e6) cap = (uint64_t)(2.0 * est + 1e6);
    }
    if (cap * sizeof(nm_dict_node) * 2 > free_b / 2) cap = free_b / 2 / (2 * sizeof(nm_dict_node));
    nm_dict_node *lists[2] = {nullptr, nullptr};
    unsigned long long *counter = nullptr;
    unsigned int *fail = nullptr;
    auto cleanup = [&]() { for (auto *p : lists) if (p) (void)hipFree(p); if (counter) (void)hipFree(counter); if (fail) (void)hipFree(fail); };
    if (cap < 1024 || hipMalloc((void **)&lists[0], cap * sizeof(nm_dict_node)) != hipSuccess || hipMalloc((void **)&lists[1], cap * sizeof(nm_dict_node)) != hipSuccess ||
        hipMalloc((void **)&counter, 8) != hipSuccess || hipMalloc((void **)&fail, 4) != hipSuccess) {
        (void)hipGetLastError();
        cleanup();
        return NM_OK;
    }
    void *table = nullptr;
    // (an error return frees the scratch and the table too: HIP_TRY alone would leave them behind)
#define DICT_TRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { nm_set_error("HIP error %d (%s) at %s:%d: %s", (int)e__, hipGetErrorString(e__), __FILE__, __LINE__, #expr); \
                                                                           cleanup(); if (table) (void)hipFree(table); return NM_E_DEVICE; } } while (0)
    nm_view v = ix->view;
    uint64_t n_nodes = 0;
    int cur = 0;
    for (uint32_t L = s; L < x; L++) {
        DICT_TRY(hipMemsetAsync(counter, 0, 8, ix->stream));
        const uint64_t n_in = L == s ? (1ULL << (2 * s)) : n_nodes;
        const uint64_t slice = 1ULL << 30;
        for (uint64_t first = 0; first < n_in; first += slice) {
            const uint64_t m = n_in - first < slice ? n_in - first : slice;
            const dim3 grid(nm_grid(m)), block(NM_BLOCK);
            const uint64_t *seed = L == s ? v.seed : nullptr;
            if (ix->big) hipLaunchKernelGGL(k_dict_expand<true>, grid, block, 0, ix->stream, v, seed, first, (const nm_dict_node *)lists[cur ^ 1], n_in, L, lists[cur], counter, cap);
            else         hipLaunchKernelGGL(k_dict_expand<false>, grid, block, 0, ix->stream, v, seed, first, (const nm_dict_node *)lists[cur ^ 1], n_in, L, lists[cur], counter, cap);
            DICT_TRY(hipGetLastError());
        }
        unsigned long long got = 0;
        DICT_TRY(hipMemcpyAsync(&got, counter, 8, hipMemcpyDeviceToHost, ix->stream));
        DICT_TRY(hipStreamSynchronize(ix->stream));
        if (got > cap) {                                            // more repeated strings than the lists hold: no dictionary
            if (nm_verbose()) fprintf(stderr, "[open] repeat dictionary: %llu nodes at length %u do not fit: none built\n", got, L + 1);
            cleanup();
            return NM_OK;
        }
        n_nodes = got;
        cur ^= 1;
    }
    const nm_dict_node *nodes = lists[cur ^ 1];
    uint32_t bits = 4;
    while ((1ULL << bits) * 4 < n_nodes) bits++;                    // <= 4 of 8 slots per bucket on average
    if (hipMalloc(&table, (128ULL << bits)) != hipSuccess) { (void)hipGetLastError(); cleanup(); return NM_OK; }
    DICT_TRY(hipMemsetAsync(table, 0xFF, (128ULL << bits), ix->stream));
    DICT_TRY(hipMemsetAsync(fail, 0, 4, ix->stream));
    if (n_nodes) hipLaunchKernelGGL(k_dict_insert, dim3(nm_grid(n_nodes)), dim3(NM_BLOCK), 0, ix->stream, nodes, n_nodes, (uint64_t *)table, bits, fail);
    unsigned int failed = 0;
    DICT_TRY(hipMemcpyAsync(&failed, fail, 4, hipMemcpyDeviceToHost, ix->stream));
    DICT_TRY(hipStreamSynchronize(ix->stream));
    cleanup();
    if (failed) { (void)hipFree(table); return NM_OK; }
#undef DICT_TRY
    ix->d_dict = table;
    ix->dict_entries = n_nodes;
    ix->device_bytes += 128ULL << bits;
    ix->view.dict = (const uint64_t *)table;
    ix->view.dict_len = x;
    ix->view.dict_bits = bits;
    if (nm_verbose()) fprintf(stderr, "[open] repeat dictionary: %llu strings of %u bases in 2^%u buckets: %.3fs\n", (unsigned long long)n_nodes, x, bits, nm_now() - td);
    return NM_OK;
}

// Range / list searches whose shortest length is below the main table's s cannot use it; they get
// a second, small table of exactly that length (built on first use, kept in the handle).
static int nm_view_for(nm_index *ix, uint32_t shortest, nm_view *v) {
    *v = ix->view;
    if (ix->view.seed_len == 0 || shortest >= ix->view.seed_len) return NM_OK;
    v->seed = nullptr;
    v->seed_len = 0;
    const uint32_t s2 = shortest > 12 ? 12 : shortest;
    if (s2 < 4) return NM_OK;
    if (ix->seed2_len != s2) {
        if (ix->d_seed2) { HIP_TRY(hipFree(ix->d_seed2)); ix->d_seed2 = nullptr; ix->device_bytes -= 8ULL << (2 * ix->seed2_len); }
        ix->seed2_len = 0;
        int rc = nm_build_seed_table(ix, s2, &ix->d_seed2);
        if (rc != NM_OK) return rc;
        ix->seed2_len = s2;
    }
    v->seed = (const uint64_t *)ix->d_seed2;
    v->seed_len = s2;
    return NM_OK;
}

extern "C" int nm_index_open(const char *index_path, int device, int seed_len_override, nm_index **out);
extern "C" int nm_index_open_budget(const char *index_path, int device, int seed_len_override, uint64_t hbm_budget_bytes, nm_index **out) {
    g_open_budget = hbm_budget_bytes;
    const int rc = nm_index_open(index_path, device, seed_len_override, out);
    g_open_budget = 0;
    return rc;
}

extern "C" int nm_index_open(const char *index_path, int device, int seed_len_override, nm_index **out) {
    if (!index_path || !out) { nm_set_error("null argument"); return NM_E_ARGUMENT; }
    *out = nullptr;
    if (device < 0) {
        nm_set_error("device %d: this engine has no CPU path; a MI355X device index (>= 0) is required", device);
        return NM_E_DEVICE;
    }
    double t_open = nm_now();
    FILE *fp = fopen(index_path, "rb");
    if (!fp) { nm_set_error("Could not load reference index from file %s", index_path); return NM_E_FILE_OPEN; }
    nm_file_header h;
    if (fread(&h, sizeof h, 1, fp) != 1 || memcmp(h.magic, NM_MAGIC, 8) != 0 || h.version != NM_FORMAT_VERSION ||
        h.header_bytes != sizeof h) {
        fclose(fp);
        nm_set_error("%s is not a newmap_amd index (format %u): rebuild it with `newmap index`", index_path, NM_FORMAT_VERSION);
        return NM_E_FILE_FORMAT;
    }
    if (h.n_rank_blocks != h.n / 64 + 1 || h.n_strand_blocks != h.n / 64 + 1 || h.n_super != (h.n >> NM_SUPER_SHIFT) + 1 ||
        h.n_super > NM_MAX_SUPER || h.off_rank != sizeof h) {
        fclose(fp);
        nm_set_error("%s: inconsistent index header", index_path);
        return NM_E_FILE_FORMAT;
    }
    int ndev = nm_device_count();
    if (device >= ndev) {
        fclose(fp);
        nm_set_error("device %d requested but %d HIP device(s) are visible", device, ndev);
        return NM_E_DEVICE;
    }
    nm_index *ix = new (std::nothrow) nm_index();
    if (!ix) { fclose(fp); nm_set_error("out of memory"); return NM_E_ALLOC; }
    ix->device = device;
    ix->h = h;
    ix->big = h.n_super > 1;
    int rc = NM_OK;
    struct Stage {                                                 // two pinned buffers the index file is uploaded through (below)
        uint8_t *buf[2] = {nullptr, nullptr};
        hipEvent_t sent[2] = {nullptr, nullptr};
        bool used[2] = {false, false};
        uint64_t turn = 0;
        bool ok = false;
        void release() {                                           // (the caller has waited for the copies out of the buffers)
            for (int i = 0; i < 2; i++) {
                if (sent[i]) (void)hipEventDestroy(sent[i]);
                if (buf[i]) (void)hipHostFree(buf[i]);
                sent[i] = nullptr; buf[i] = nullptr;
            }
            ok = false;
        }
        ~Stage() { release(); }
    } stage;
    auto fail = [&](int code) {
        if (ix->stream) (void)hipStreamSynchronize(ix->stream);   // (a copy out of a staging buffer may be on its way)
        stage.release();
        fclose(fp);
        nm_index_close(ix);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) { nm_set_error("hipSetDevice(%d) failed", device); return fail(NM_E_DEVICE); }
    if (hipStreamCreate(&ix->stream) != hipSuccess) { nm_set_error("hipStreamCreate failed"); return fail(NM_E_DEVICE); }
    ix->lanes[0].owner = ix->stream;
    NM_PHASE(t_open, "header + device initialisation");

    const uint64_t rank_bytes = h.n_rank_blocks * sizeof(nm_rank_block);
    const uint64_t strand_bytes = h.n_strand_blocks * sizeof(nm_strand_block);
    const uint64_t sep_bytes = (h.n_sep ? h.n_sep : 1) * sizeof(uint64_t);
    std::vector<uint64_t> superC(h.n_super * 4);
    uint64_t C[4];
    C[0] = h.n_sep;
    for (int c = 1; c < 4; c++) C[c] = C[c - 1] + h.base_count[c - 1];
    for (uint64_t j = 0; j < h.n_super; j++)
        for (int c = 0; c < 4; c++) superC[j * 4 + c] = C[c] + h.super_cnt[j][c];

    // stage through two pinned buffers: the file is read once, sequentially, and the read of a chunk (page cache -> pinned)
    // overlaps the DMA of the chunk before it (a pageable buffer made every chunk wait for its own copy: 4.6 GB of a human
    // index in ~0.25 s of the one-shot CLI's 1.2 s)
    const uint64_t chunk = 64ULL << 20;
    stage.ok = hipHostMalloc((void **)&stage.buf[0], chunk, hipHostMallocDefault) == hipSuccess &&
               hipHostMalloc((void **)&stage.buf[1], chunk, hipHostMallocDefault) == hipSuccess &&
               hipEventCreateWithFlags(&stage.sent[0], hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&stage.sent[1], hipEventDisableTiming) == hipSuccess;
    if (!stage.ok) (void)hipGetLastError();
    auto upload = [&](void **dptr, uint64_t off, uint64_t bytes) -> int {
        if (hipMalloc(dptr, (bytes ? bytes : 8) + 64) != hipSuccess) { nm_set_error("hipMalloc of %llu bytes failed", (unsigned long long)bytes); return NM_E_ALLOC; }   // (+ 64: the sweep reads 32 LCP bytes at a time)
        ix->device_bytes += bytes;
        if (!stage.ok) {                                               // (no pinned memory to be had: one pageable buffer)
            if (fseeko(fp, (off_t)off, SEEK_SET) != 0) { nm_set_error("seek failed in %s", index_path); return NM_E_FILE_FORMAT; }
            std::vector<uint8_t> buf((size_t)(bytes < chunk ? bytes : chunk));
            for (uint64_t done = 0; done < bytes;) {
                const uint64_t m = bytes - done < chunk ? bytes - done : chunk;
                if (fread(buf.data(), 1, (size_t)m, fp) != m) { nm_set_error("%s is truncated", index_path); return NM_E_FILE_FORMAT; }
                if (hipMemcpy((uint8_t *)*dptr + done, buf.data(), m, hipMemcpyHostToDevice) != hipSuccess) { nm_set_error("hipMemcpy to device failed"); return NM_E_DEVICE; }
                done += m;
            }
            return NM_OK;
        }
        const int fd = fileno(fp);
        for (uint64_t done = 0; done < bytes;) {
            const uint64_t m = bytes - done < chunk ? bytes - done : chunk;
            const int b = (int)(stage.turn++ & 1u);
            if (stage.used[b] && hipEventSynchronize(stage.sent[b]) != hipSuccess) { nm_set_error("waiting for an upload failed"); return NM_E_DEVICE; }
            // (one thread copies out of the page cache at ~17 GB/s, a third of what the bus takes: four read a chunk together)
            std::atomic<bool> short_read{false};
            auto read_part = [&](uint64_t a, uint64_t e) {
                for (uint64_t got = a; got < e;) {
                    const ssize_t r = pread(fd, stage.buf[b] + got, (size_t)(e - got), (off_t)(off + done + got));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) { short_read = true; return; }
                    got += (uint64_t)r;
                }
            };
            constexpr int kReaders = 4;
            const uint64_t part = ((m + kReaders - 1) / kReaders + 4095) & ~4095ULL;
            std::thread helpers[kReaders - 1];
            int n_helpers = 0;
            for (int t = 1; t < kReaders && (uint64_t)t * part < m; t++)
                helpers[n_helpers++] = std::thread(read_part, (uint64_t)t * part, ((uint64_t)t + 1) * part < m ? ((uint64_t)t + 1) * part : m);
            read_part(0, part < m ? part : m);
            for (int t = 0; t < n_helpers; t++) helpers[t].join();
            if (short_read) { nm_set_error("%s is truncated", index_path); return NM_E_FILE_FORMAT; }
            if (hipMemcpyAsync((uint8_t *)*dptr + done, stage.buf[b], m, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
                hipEventRecord(stage.sent[b], ix->stream) != hipSuccess) { nm_set_error("copy to device failed"); return NM_E_DEVICE; }
            stage.used[b] = true;
            done += m;
        }
        return NM_OK;
    };
    if ((rc = upload(&ix->d_rank, h.off_rank, rank_bytes)) != NM_OK) return fail(rc);
    if ((rc = upload(&ix->d_strand, h.off_strand, strand_bytes)) != NM_OK) return fail(rc);
    if ((rc = upload(&ix->d_sep, h.off_sep, h.n_sep * sizeof(uint64_t))) != NM_OK) return fail(rc);
    // the LCP bytes (optional section; resident handles only: the one-shot CLI's run is bound by the host, and n more bytes
    // to read and upload are not earned back by one search).  NEWMAP_AMD_LCP=0: not loaded (A/B)
    {
        const char *lcp_env = getenv("NEWMAP_AMD_LCP");
        if (h.off_lcp && seed_len_override == -2 && !(lcp_env && lcp_env[0] == '0') &&
            (rc = upload(&ix->d_lcp, h.off_lcp, h.n + 1)) != NM_OK) return fail(rc);
    }
    if (hipStreamSynchronize(ix->stream) != hipSuccess) { nm_set_error("upload of %s failed", index_path); return fail(NM_E_DEVICE); }
    stage.release();
    (void)sep_bytes;
    if (hipMalloc(&ix->d_super, superC.size() * sizeof(uint64_t)) != hipSuccess ||
        hipMemcpy(ix->d_super, superC.data(), superC.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
        nm_set_error("could not upload the superblock table");
        return fail(NM_E_DEVICE);
    }
    {   // the record list (format 2) and the fingerprint tables
        ix->records.resize(h.n_records);
        if (h.n_records && (fseeko(fp, (off_t)h.off_records, SEEK_SET) != 0 ||
                            fread(ix->records.data(), sizeof(nm_record_entry), h.n_records, fp) != h.n_records)) {
            nm_set_error("%s is truncated (record list)", index_path);
            return fail(NM_E_FILE_FORMAT);
        }
        std::sort(ix->records.begin(), ix->records.end(), [](const nm_record_entry &a, const nm_record_entry &b) {
            return a.length != b.length ? a.length < b.length : a.hash < b.hash; });
        std::vector<uint64_t> tab(NM_HASH_TAB_WORDS);
        nm_hash_fill_tables(tab.data());
        if (hipMalloc(&ix->d_hash_tab, tab.size() * sizeof(uint64_t)) != hipSuccess ||
            hipMemcpy(ix->d_hash_tab, tab.data(), tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
            nm_set_error("could not upload the fingerprint tables");
            return fail(NM_E_DEVICE);
        }
    }
    fclose(fp);
    fp = nullptr;
    NM_PHASE(t_open, "index file read + upload");

    nm_view &v = ix->view;
    v.rank = (const nm_rank_block *)ix->d_rank;
    v.strand = (const nm_strand_block *)ix->d_strand;
    v.sep = (const uint64_t *)ix->d_sep;
    v.seed = nullptr;
    v.superC = (const uint64_t *)ix->d_super;
    v.n = h.n;
    v.n_sep = h.n_sep;
    for (int c = 0; c < 4; c++) v.C[c] = C[c];
    v.seed_len = 0;
    v.n_super = (uint32_t)h.n_super;
    v.seed_policy = 0;
    v.lfb = nullptr;
    v.quad = nullptr;
    v.quad_m = 0;
    v.quad2 = nullptr;
    v.quad2_m = 0;
    v.hash_tab = (const uint64_t *)ix->d_hash_tab;
    v.dict = nullptr;
    v.dict_len = v.dict_bits = 0;
    v.lf2 = nullptr;
    v.lcp = (const uint8_t *)ix->d_lcp;

    if (seed_len_override < -1 && h.n >= 2) {
        const char *off = getenv("NEWMAP_AMD_LF_BLOCKS");
        if (!(off && off[0] == '0')) {
            rc = nm_build_lf_blocks(ix);
            if (rc != NM_OK) { nm_index_close(ix); return rc; }
            NM_PHASE(t_open, "LF blocks");
        }
    }
    uint32_t s = seed_len_override == -1 ? h.seed_len
               : (seed_len_override < -1 ? nm_auto_seed_len(ix) : (uint32_t)seed_len_override);
    if (s > 16) s = 16;
    // -3: automatic with small tables (seed <= 15, quad cores <= 13 + a table with shorter cores: 20 GB at most).  A one-shot run never earns
    // back what the large tables cost to allocate: hipMalloc of more than ~40 GB waits 3 - 5 s for the driver to
    // clear the memory (measured), the large tables save ~1.5 ps per position.
    const bool small_tables = seed_len_override == -3;
    if (small_tables && s > 15) s = 15;
    // automatic sizing: the quad table, cut from the seed-table level of its core length.  Core length:
    // NEWMAP_AMD_QUAD_M (0 = none), else nm_auto_quad_len.
    uint32_t quad_m = 0;
    if (seed_len_override < -1 && s >= 8) {
        quad_m = nm_auto_quad_len(ix, s);
        {   // cost-aware: cores longer than this genome needs buy nothing -- with windows of ceil(log4(200 n)) bases one window
            // in two hundred is repeated, and what they leave open is a per-mille of the positions.  100 Mbp: cores of 14
            // (34 GB, open in ~1 s) instead of 15 (137 GB, ~4 s); 3 Gbp: the HBM is the limit as before (15)
            uint32_t w = 1;
            while (w < 32 && (double)(1ULL << (2 * w)) < 200.0 * (double)h.n) w++;
            const uint32_t enough = w > NM_QUAD_EXT + 8 ? w - NM_QUAD_EXT : 8;
            if (!getenv("NEWMAP_AMD_QUAD_FULL") && quad_m > enough) quad_m = enough;
        }
        if (small_tables && quad_m > 13) quad_m = 13;
        if (const char *q = getenv("NEWMAP_AMD_QUAD_M")) quad_m = (uint32_t)atoi(q);
        if (quad_m > s) quad_m = s;
        if (quad_m && quad_m < 8) quad_m = 8;              // the level-wise build starts at length 8
    }
    // a second quad table with SHORT cores (larger groups per table line, nm_core.h "sites"): windows of
    // ceil(log4(20 n)) bases -- about one in twenty repeated -- when that is shorter than the first table's and the
    // table stays below 9 GB (cores <= 13; 14 on large genomes, see below); NEWMAP_AMD_QUAD_SMALL_M overrides (0 = none)
    uint32_t quad_small_m = 0;
    if (quad_m) {
        uint32_t w1 = 1;
        while (w1 < 32 && (double)(1ULL << (2 * w1)) < 20.0 * (double)h.n) w1++;
        quad_small_m = w1 > NM_QUAD_EXT + 8 ? w1 - NM_QUAD_EXT : 8;
        if (quad_small_m > 13) {                             // capped: worth its memory only while most of its windows still occur once
            auto repeated = [&](uint32_t m) { return 1.0 - exp(-(double)h.n / pow(4.0, (double)(m + NM_QUAD_EXT))); };
            quad_small_m = 13;
            if (repeated(13) > 0.15) {
                // genomes of several Gbp: cores of 14 (34 GB) for a resident handle, when the memory left after the seed
                // table and the first quad table holds that twice (3.09 Gbp, 20:200: 7 positions per line instead of 6)
                quad_small_m = 0;
                size_t free_b = 0, total_b = 0;
                const uint64_t first = (8ULL << (2 * s)) + (128ULL << (2 * quad_m)), want = 128ULL << 28;
                if (!small_tables && quad_m > 14 && repeated(14) <= 0.15 && nm_free_hbm(ix, &free_b, &total_b) == hipSuccess &&
                    free_b > first && want <= (free_b - first) / 2)
                    quad_small_m = 14;
            }
        }
        if (const char *q = getenv("NEWMAP_AMD_QUAD_SMALL_M")) quad_small_m = (uint32_t)atoi(q);
        if (quad_small_m && quad_small_m < 8) quad_small_m = 8;
        if (quad_small_m >= quad_m) quad_small_m = 0;
    }
    rc = nm_build_seed(ix, s, quad_m, quad_small_m);
    if (rc != NM_OK) { nm_index_close(ix); return rc; }
    // (not with the small tables of the one-shot CLI: its run is bound by the host, and the dictionary costs ~0.5 s to build)
    if (seed_len_override == -2 && (rc = nm_build_dict(ix)) != NM_OK) { nm_index_close(ix); return rc; }
    if (seed_len_override == -2 && (rc = nm_build_lf2(ix)) != NM_OK) { nm_index_close(ix); return rc; }
    NM_PHASE(t_open, "seed and quad tables, repeat dictionary, two-base LF blocks");
    if (const char *cm = getenv("NEWMAP_AMD_COARSE_MIN")) ix->coarse_min = strtoull(cm, nullptr, 10);
    if (const char *cm = getenv("NEWMAP_AMD_COARSE")) ix->coarse_mode = atoi(cm);
    if (const char *cs = getenv("NEWMAP_AMD_COARSE_STRIDE")) { const int v = atoi(cs); if (v == 128 || v == 256 || v == 512) ix->coarse_stride = (uint32_t)v; }
    if (const char *pb = getenv("NEWMAP_AMD_PROBES_BESIDE")) ix->probes_beside = pb[0] != '0';
    if (const char *pr = getenv("NEWMAP_AMD_PERIODIC")) ix->periodic_runs = pr[0] != '0';
    if (const char *sw = getenv("NEWMAP_AMD_SWEEP")) { const int v = atoi(sw); if (v >= 0 && v <= 2) ix->sweep = v; }
#ifdef NM_MEASURE
    if (const char *sb = getenv("NEWMAP_AMD_SITES_BLOCKS_PER_CU")) ix->sites_blocks_per_cu = atoi(sb);      // occupancy cap of k_sites (LDS padding)
#endif
    if (const char *sd = getenv("NEWMAP_AMD_SITE_D")) { ix->site_d_cap = (uint32_t)atoi(sd); if (ix->site_d_cap > NM_SITE_MAX_D) ix->site_d_cap = NM_SITE_MAX_D; }
    if (hipHostMalloc((void **)&ix->h_repeats_seen, 64, hipHostMallocMapped) == hipSuccess) {
        ix->h_repeats_seen[0] = ix->h_repeats_seen[1] = 0;     // [0]: long repeats met (fine probes), [1]: open positions met (k_resolve)
        if (hipHostGetDevicePointer((void **)&ix->d_repeats_seen, ix->h_repeats_seen, 0) != hipSuccess ||
            hipMalloc((void **)&ix->d_seen_latch, 64) != hipSuccess || hipMemset(ix->d_seen_latch, 0, 64) != hipSuccess) {
            (void)hipGetLastError();
            ix->d_repeats_seen = nullptr;
        }
    } else {
        (void)hipGetLastError();
        ix->h_repeats_seen = nullptr;
    }
    rc = nm_grow(ix->status, NM_STATUS_WORDS * sizeof(uint64_t));
    if (rc == NM_OK) rc = nm_lane_ready(ix->lanes[0]);
    if (rc != NM_OK) { nm_index_close(ix); return rc; }
    *out = ix;
    return NM_OK;
}

extern "C" void nm_index_close(nm_index *ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    (void)hipDeviceSynchronize();                              // launches on caller streams and side streams included
    void *ptrs[] = {ix->d_rank, ix->d_strand, ix->d_sep, ix->d_seed, ix->d_seed2, ix->d_quad, ix->d_quad_small, ix->d_lfb, ix->d_super, ix->d_hash_tab, ix->d_dict, ix->d_lf2, ix->d_lcp, ix->seq.p,
                    ix->out.p, ix->status.p, ix->starts.p, ix->lens.p};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (nm_lane &L : ix->lanes) {
        for (void *p : {L.enc.p, L.ks.p, L.work.p, L.settled.p, L.coarse.p, L.need.p, L.need2.p, L.hashp.p, L.open_list.p})
            if (p) (void)hipFree(p);
        if (L.side) (void)hipStreamDestroy(L.side);
        for (hipEvent_t e : {L.ev_fork, L.ev_join, L.ev_last})
            if (e) (void)hipEventDestroy(e);
    }
    if (ix->h_repeats_seen) (void)hipHostFree(ix->h_repeats_seen);
    if (ix->d_seen_latch) (void)hipFree(ix->d_seen_latch);
    for (auto &pool : ix->ev_pool) for (hipEvent_t e : pool) (void)hipEventDestroy(e);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

extern "C" uint64_t nm_index_info(const nm_index *ix, int what) {
    if (!ix) return 0;
    switch (what) {
        case 0: return ix->h.n;
        case 1: return ix->h.n_fwd;
        case 2: return ix->h.n_sep;
        case 3: return ix->h.n_records;
        case 4: return ix->h.raw_bases;
        case 5: return ix->view.seed_len;
        case 6: return ix->device_bytes;
        case 7: return ix->h.sa_ratio;
        case 8: return (uint64_t)ix->last_kernel;
        case 9: return 0;                                  // (pair table: removed)
        case 10: return (uint64_t)ix->device;
        case 11: return ix->view.lfb ? 1 : 0;
        case 12: return 0;                                 // (two-step rank blocks: removed)
        case 13: return ix->repeat_probes ? 1 : 0;
        case 18: return ix->view.quad_m;
        case 19: return ix->quad_small_m;
        case 20: return ix->last_site_m;
        case 21: return ix->last_fingerprint;
        case 22: return ix->initial_len;
        case 23: return ix->guard_segments;
        case 26: return ix->view.lf2 ? 1 : 0;
        case 35: return ix->d_lcp != nullptr;               // LCP bytes resident
        case 24: return ix->view.dict_len;
        case 25: return ix->dict_entries;
        case 27: case 28: case 29: case 30: case 31: case 32: case 33: case 34: {      // k_open_words of the last launch: words / open positions by class
            unsigned long long v = 0;
            if (hipSetDevice(ix->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 0;
            if (hipMemcpy(&v, (const unsigned long long *)ix->cur->work.p + NM_WORK_LIST + (what - 27), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
            return v;
        }
        case 14: case 15: case 16: case 17: {              // probe tally of the last range-mode launch
            unsigned long long v = 0;
            if (hipSetDevice(ix->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 0;
            if (hipMemcpy(&v, (const unsigned long long *)ix->cur->work.p + 1 + (what - 14), sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return 0;
            return v;
        }
        default: return 0;
    }
}

extern "C" int nm_set_option(nm_index *ix, int option, int64_t value) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (option == NM_OPT_COUNT_STEPS) { ix->count_steps = value != 0; return NM_OK; }
    if (option == NM_OPT_TIMING) { ix->timing = value != 0; for (size_t &u : ix->ev_used) u = 0; return NM_OK; }
    if (option == NM_OPT_LF_BLOCKS) {      // A/B: LF steps read the 16-byte LF entries (if built) or the packed blocks
        ix->view.lfb = value ? (const nm_lf_entry *)ix->d_lfb : nullptr;
        return NM_OK;
    }
    if (option == NM_OPT_SEED_POLICY) {
        if (value < 0 || (value & 0xFF) > 2 || value > 0x3FFF) { nm_set_error("seed policy must be 0, 1 or 2 (+ A/B bits 0x800 .. 0x2000)"); return NM_E_ARGUMENT; }
#ifndef NM_MEASURE
        if (value & 0x700) { nm_set_error("seed policy bits 0x100 / 0x200 cut work out of the kernels and give wrong results: measurement build only (make -C newmap_amd/csrc measure)"); return NM_E_ARGUMENT; }
#endif
        ix->view.seed_policy = (uint32_t)value;
        return NM_OK;
    }
    if (option == NM_OPT_FORCE_BIG) {      // tests: run the >2^31-position code path on a small index
        ix->big = value != 0 || ix->h.n_super > 1;
        return NM_OK;
    }
    if (option == NM_OPT_LIST_VIA_RANGE) {
        ix->list_via_range = value != 0;
        return NM_OK;
    }
    if (option == NM_OPT_REPEAT_PROBES) {
        ix->repeat_probes = value != 0;
        return NM_OK;
    }
    if (option == NM_OPT_SITE_D) {         // measurement / tests: cap on d = kmin - window of the sites (a group = d + 5 positions)
        if (value < 0 || value > (int64_t)NM_SITE_MAX_D) { nm_set_error("site d cap must be 0..%u", NM_SITE_MAX_D); return NM_E_ARGUMENT; }
        ix->site_d_cap = (uint32_t)value;
        return NM_OK;
    }
    if (option == NM_OPT_SITE_TABLE) {
        if (value < 0 || value > 2) { nm_set_error("site table must be 0 (pick per launch), 1 (long cores) or 2 (short cores)"); return NM_E_ARGUMENT; }
        ix->site_table = (int)value;
        return NM_OK;
    }
    if (option == NM_OPT_LF2) { ix->view.lf2 = value ? (const nm_lf_entry *)ix->d_lf2 : nullptr; return NM_OK; }
    if (option == NM_OPT_LCP) { ix->view.lcp = value ? (const uint8_t *)ix->d_lcp : nullptr; return NM_OK; }
    if (option == NM_OPT_SWEEP) {
        if (value < 0 || value > 2) { nm_set_error("sweep must be 0 (k_resolve only), 1 (once the handle has met open positions) or 2 (always)"); return NM_E_ARGUMENT; }
        ix->sweep = (int)value;
        return NM_OK;
    }
    if (option == NM_OPT_SEGMENT_GUARD) { ix->segment_guard = value != 0; return NM_OK; }
    if (option == NM_OPT_INITIAL_LENGTH) {
        if (value < 0 || value > 0xFFFFFFFFLL) { nm_set_error("initial search length out of range"); return NM_E_ARGUMENT; }
        ix->initial_len = (uint32_t)value;
        return NM_OK;
    }
    if (option == NM_OPT_KERNEL) {
        if (value != 0 && value != 1 && value != 5) { nm_set_error("kernel version must be 0 (automatic), 1 (one lane per position) or 5 (sites)"); return NM_E_ARGUMENT; }
        ix->kernel_version = (int)value;
        return NM_OK;
    }
    nm_set_error("unknown option %d", option);
    return NM_E_ARGUMENT;
}

extern "C" int nm_timing_read_kind(nm_index *ix, int kind, uint64_t *n_launches, double *total_ms, double *max_ms) {
    if (!ix) { nm_set_error("null handle"); return NM_E_ARGUMENT; }
    if (kind < 0 || kind >= NM_TIMING_KINDS) { nm_set_error("timing kind must be 0 (search kernel), 1 (all kernels of a segment), 2 (coarse probes), 3 (fine probes), 4 (k_open_words + k_sweep + k_resolve) or 5 (k_sweep)"); return NM_E_ARGUMENT; }
    HIP_TRY(hipSetDevice(ix->device));
    double total = 0.0, mx = 0.0;
    std::vector<hipEvent_t> &pool = ix->ev_pool[kind];
    for (size_t i = 0; i + 1 < ix->ev_used[kind]; i += 2) {
        HIP_TRY(hipEventSynchronize(pool[i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pool[i], pool[i + 1]));
        total += ms;
        if (ms > mx) mx = ms;
    }
    if (n_launches) *n_launches = ix->ev_used[kind] / 2;
    if (total_ms) *total_ms = total;
    if (max_ms) *max_ms = mx;
    ix->ev_used[kind] = 0;
    return NM_OK;
}

extern "C" int nm_timing_read(nm_index *ix, uint64_t *n_launches, double *total_ms, double *max_ms) {
    return nm_timing_read_kind(ix, 0, n_launches, total_ms, max_ms);
}

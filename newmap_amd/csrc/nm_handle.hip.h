// nm_handle.hip.h -- the index handle: device buffers, lanes (launch scratch per caller stream), timing events (part of
// nm_engine.hip).
// ------------------------------------------------------------------------------ host side ---

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            nm_set_error("HIP error %d (%s) at %s:%d: %s", (int)e__, hipGetErrorString(e__),  \
                         __FILE__, __LINE__, #expr);                                          \
            return NM_E_DEVICE;                                                               \
        }                                                                                     \
    } while (0)

struct nm_buffer {
    void *p = nullptr;
    uint64_t bytes = 0;
};

#define NM_LANES 6
#define NM_TIMING_KINDS 6
struct nm_lane {
    hipStream_t owner = nullptr;          // the stream whose launches use this scratch
    bool ready = false;                   // side stream and events exist
    uint64_t tick = 0;                    // last use (LRU)
    hipStream_t side = nullptr;           // repeat probes of repeat-rich input run here, beside k_sites (launch_sites)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_last = nullptr;         // end of the lane's last call on its owner's stream
    nm_buffer enc, ks, work, settled, coarse, need, need2, hashp, open_list;   // grown on demand
    uint64_t enc_words = 0;               // words written by the last nm_encode
};

struct nm_index {
    int device = 0;
    nm_file_header h;
    nm_view view;
    bool big = false;
    void *d_rank = nullptr, *d_strand = nullptr, *d_sep = nullptr, *d_seed = nullptr, *d_super = nullptr;
    void *d_seed2 = nullptr;              // small secondary seed table (nm_view_for)
    uint32_t seed2_len = 0;
    void *d_quad = nullptr;               // quad table (k_sites), cores as long as the memory allows
    void *d_quad_small = nullptr;         // a second one with shorter cores: larger groups per line on small genomes
    uint32_t quad_small_m = 0;
    void *d_lfb = nullptr;                // LF blocks
    void *d_dict = nullptr;               // repeat dictionary (nm_core.h)
    void *d_lf2 = nullptr;                // two-base LF blocks
    void *d_lcp = nullptr;                // LCP bytes of the index file (nm_format.h), resident handles only
    uint64_t dict_entries = 0;
    void *d_hash_tab = nullptr;           // tables of the record fingerprint (nm_hash.h)
    std::vector<nm_record_entry> records; // (length, fingerprint) of the indexed records, sorted
    uint64_t device_bytes = 0;
    hipStream_t stream = nullptr;
    // Launch scratch comes in LANES, one per stream the caller launches on: segments given on different streams have
    // their own encoded words, bitmaps and counters and may overlap on the GPU (a 10 M-position launch leaves most of
    // the chip idle while its last blocks drain and its three small kernels run).  Lane 0 belongs to the handle's own
    // stream (host-buffer entry points); a caller stream keeps its lane until more than NM_LANES - 1 streams are in use,
    // then the least recently used lane changes hands behind its `ev_last` (nm_lane_for).
    nm_lane lanes[NM_LANES];
    nm_lane *cur = &lanes[0];             // lane of the call in progress / of the last call (nm_index_info 14..17)
    uint64_t lane_tick = 0;
    // scratch of the host-buffer entry points (they run on `stream`, one call at a time)
    nm_buffer seq, out, status, starts, lens;
    uint64_t coarse_min = 32ull << 20;    // launches of at least this many positions also run the coarse probes (NEWMAP_AMD_COARSE_MIN) ...
    uint32_t coarse_stride = NM_COARSE_STRIDE;   // positions per coarse probe (NEWMAP_AMD_COARSE_STRIDE: 128, 256, 512)
    int coarse_mode = 1;                  // ... 1: once an earlier launch has met long repeats, 2: always, 0: never (NEWMAP_AMD_COARSE)
    uint32_t *h_repeats_seen = nullptr;   // pinned word the fine probes set; d_repeats_seen = its device address
    uint32_t *d_repeats_seen = nullptr;
    uint32_t *d_seen_latch = nullptr;     // device-side copy of the flag
    bool list_via_range = true;           // list mode with one length runs on the range kernels (NM_OPT_LIST_VIA_RANGE, A/B)
    bool repeat_probes = true;            // k_repeat_probe before the both-strand range kernels (NM_OPT_REPEAT_PROBES)
    int kernel_version = 0;               // 0 = automatic (sites when the quad table applies, else 1); 1 / 5 force a kernel
    uint32_t last_site_m = 0;             // core length of the table the sites of the last launch read (nm_index_info 20)
    bool probes_beside = true;            // NEWMAP_AMD_PROBES_BESIDE=0: the probes always follow k_sites on its stream (A/B)
    int sites_blocks_per_cu = 0;          // NEWMAP_AMD_SITES_BLOCKS_PER_CU (0 = as many as fit)
    bool periodic_runs = true;            // NEWMAP_AMD_PERIODIC=0: the coarse probes walk every stride (A/B) instead of one walk per tandem run
    int site_table = 0;                   // measurement knob (NM_OPT_SITE_TABLE): 0 = pick per launch, 1 = long cores, 2 = short cores
    uint32_t site_d_cap = NM_SITE_MAX_D;  // measurement knob (NEWMAP_AMD_SITE_D): cap on d = kmin - window of the sites
    int sweep = 1;                        // NM_OPT_SWEEP: 1 = open positions are offered to k_sweep once the handle has met any (default), 2 = always, 0 = k_resolve only (A/B)
    bool count_steps = false;
    int last_kernel = 0;                  // which range kernel the last launch used (nm_index_info 8)
    uint64_t guard_segments = 0;          // segments that went through the exact guard (nm_index_info 23; tests)
    bool segment_guard = true;            // NM_OPT_SEGMENT_GUARD: the host-buffer segment calls run the exact guard themselves
    uint32_t initial_len = 0;             // --initial-search-length (NM_OPT_INITIAL_LENGTH): shapes the reference's probe schedule, hence the guard
    uint64_t last_fingerprint = 0;        // status[NM_STATUS_HASH] of the last host-buffer segment call (nm_index_info 21)
    // NM_OPT_TIMING: HIP events on the launch stream, NM_TIMING_KINDS kinds of start/stop pairs:
    // kind 0 around the dominant search kernel of a segment (k_sites / k_min_unique / k_fixed_k), kind 1 around ALL the
    // kernels of the segment (encode pass, sites, probes, resolve), kinds 2 / 3 / 4 around the coarse probes, the fine
    // probes and the finishing stage (k_open_words + k_sweep + k_resolve), kind 5 around k_sweep alone (each on the stream it is
    // launched on: the probes may run on the lane's side stream)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool[NM_TIMING_KINDS];   // start/stop pairs, reused
    size_t ev_used[NM_TIMING_KINDS] = {0, 0, 0, 0, 0, 0};  // events consumed since the last read
};

struct nm_timed {                         // records start on construction, stop on destruction
    nm_index *ix; hipStream_t st; hipEvent_t stop = nullptr;
    nm_timed(nm_index *ix_, hipStream_t st_, int kind = 0) : ix(ix_), st(st_) {
        if (!ix->timing) return;
        std::vector<hipEvent_t> &pool = ix->ev_pool[kind];
        size_t &used = ix->ev_used[kind];
        if (used + 2 > pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            pool.push_back(a); pool.push_back(b);
        }
        (void)hipEventRecord(pool[used], st);
        stop = pool[used + 1];
        used += 2;
    }
    ~nm_timed() { if (stop) (void)hipEventRecord(stop, st); }
};

static int nm_grow(nm_buffer &b, uint64_t bytes) {
    if (bytes <= b.bytes && b.p) return NM_OK;
    if (b.p) { HIP_TRY(hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    uint64_t want = bytes + bytes / 8 + 4096;
    HIP_TRY(hipMalloc(&b.p, want));
    b.bytes = want;
    return NM_OK;
}

// side stream, events and counters of a lane (once)
static int nm_lane_ready(nm_lane &L) {
    if (L.ready) return NM_OK;
    if (hipStreamCreateWithFlags(&L.side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&L.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&L.ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();                               // (without them the probes simply follow k_sites on one stream)
        L.side = nullptr;
    }
    HIP_TRY(hipEventCreateWithFlags(&L.ev_last, hipEventDisableTiming));
    int rc = nm_grow(L.work, NM_WORK_WORDS * sizeof(unsigned long long));
    if (rc != NM_OK) return rc;
    L.ready = true;
    return NM_OK;
}

// the lane of a launch on stream `st` becomes ix->cur.  A stream keeps its lane; a new stream takes a free lane, or the
// least recently used one of lanes 1.. after waiting (on the device) for that lane's last call.
static int nm_lane_for(nm_index *ix, hipStream_t st) {
    nm_lane *pick = nullptr;
    for (nm_lane &L : ix->lanes)
        if (L.owner == st && (L.ready || &L == &ix->lanes[0])) { pick = &L; break; }
    if (!pick) {
        for (int i = 1; i < NM_LANES && !pick; i++)
            if (!ix->lanes[i].owner) pick = &ix->lanes[i];
        if (!pick) {
            pick = &ix->lanes[1];
            for (int i = 2; i < NM_LANES; i++)
                if (ix->lanes[i].tick < pick->tick) pick = &ix->lanes[i];
            HIP_TRY(hipStreamWaitEvent(st, pick->ev_last, 0));
        }
        pick->owner = st;
    }
    int rc = nm_lane_ready(*pick);
    if (rc != NM_OK) return rc;
    pick->tick = ++ix->lane_tick;
    ix->cur = pick;
    return NM_OK;
}

// a caller that is about to destroy a stream gives its lane back: the scratch stays for the next stream that needs one
extern "C" int nm_stream_release(nm_index *ix, void *stream) {
    if (!ix || !stream) return NM_OK;
    for (int i = 1; i < NM_LANES; i++) {
        nm_lane &L = ix->lanes[i];
        if (L.owner != (hipStream_t)stream) continue;
        HIP_TRY(hipSetDevice(ix->device));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        if (L.side) HIP_TRY(hipStreamSynchronize(L.side));
        L.owner = nullptr;
        L.tick = 0;
        if (ix->cur == &L) ix->cur = &ix->lanes[0];
    }
    return NM_OK;
}

// end of a call: whoever takes the lane over later waits for this point of the owner's stream
static int nm_lane_done(nm_index *ix, hipStream_t st) {
    if (ix->cur != &ix->lanes[0]) HIP_TRY(hipEventRecord(ix->cur->ev_last, st));
    return NM_OK;
}

static inline unsigned nm_grid(uint64_t items) { return (unsigned)((items + NM_BLOCK - 1) / NM_BLOCK); }

extern "C" int nm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

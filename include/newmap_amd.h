/*
 * newmap_amd.h -- C-ABI of the MI355X-native engine behind newmap's `search` hot path.
 *
 * This is the drop-in boundary: plain pointers, sizes and int return codes, no exceptions, no
 * torch / Python types.  It replaces the two CPython extension modules through which the
 * reference reaches AvxWindowFmIndex (file:line relative to the reference checkout):
 *
 *   newmap._c_newmap_generate_index.generate_fm_index   src/newmap-generate-index.c:11-58,82-104
 *   newmap._c_newmap_count_kmers.count_kmers            src/newmap-count.c:28-89
 *   newmap._c_newmap_count_kmers.count_kmers_from_sequence
 *                                                        src/newmap-count.c:91-206
 * and hoists the per-segment search loop that sits on top of that seam
 *   newmap.search.binary_search / linear_search          newmap/search.py:383-548, 551-644
 *   (mask :744-766, upper bound :769-882, strand sum + zero guard :647-724)
 * into one fused device launch per segment.
 *
 * Ownership: every buffer is caller-owned.  `nm_index*` is an opaque handle that owns the
 * device-resident index (uploaded ONCE, unlike src/newmap-count.c:135-136 which re-reads the
 * index file on every call).  One handle per device; calls on one handle are serialised by the
 * caller.  All functions return NM_OK (0) or a positive NM_E* code; nm_last_error() gives the
 * message of the last failure on the calling thread.
 */
#ifndef NEWMAP_AMD_H
#define NEWMAP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nm_index nm_index;

enum {
    NM_OK = 0,
    NM_E_FILE_OPEN = 1,      /* FileNotFoundError   (AwFmFileOpenFail,      newmap-generate-index.c:32-36) */
    NM_E_ALLOC = 2,          /* MemoryError         (AwFmAllocationFailure, :38-42)                        */
    NM_E_FILE_EXISTS = 3,    /* FileExistsError     (AwFmFileAlreadyExists, :44-48) -- never returned:
                                like the reference at its pinned library version, build overwrites     */
    NM_E_FILE_WRITE = 4,     /* OSError             (AwFmFileWriteFail,     :50-54)                        */
    NM_E_FILE_FORMAT = 5,    /* OSError: not an index written by nm_index_build (newmap-count.c:13-16) */
    NM_E_ARGUMENT = 6,       /* ValueError / IndexError raised by the Python wrapper                    */
    NM_E_DEVICE = 7,         /* RuntimeError: HIP failure, no device, kernel fault                      */
    NM_E_KMER_NOT_FOUND = 8, /* RuntimeError "k-mer was not found in the index" (search.py:699-722)     */
    NM_E_TOO_LARGE = 9       /* text too large for this build                                           */
};

const char *nm_last_error(void);
const char *nm_version(void);

/* ---------------------------------------------------------------------------------- index ---
 * Host-side build: FASTA -> both-strand text -> suffix array -> BWT -> rank blocks -> file.
 * Replaces generate_fm_index(fasta, index, compression_ratio, seed_length)
 * (src/newmap-generate-index.c:11-58).  `sa_ratio` is accepted and recorded for interface parity
 * (counting never touches a sampled suffix array); `seed_len` is the default length of the
 * device seed table built at nm_index_open time. */
int nm_index_build(const char *fasta_path, const char *index_path, uint8_t sa_ratio, uint8_t seed_len);

/* The same build with the suffix array computed on `device` (prefix doubling over rocPRIM radix sorts,
 * csrc/nm_build_device.hip); the index file is byte-identical to nm_index_build's.  Texts of 2^31
 * symbols or more (genomes beyond ~1.07 Gbp) take a bucketed 64-bit variant that also gathers the BWT
 * on the device (3.09 Gbp: 10 s instead of 146 s); if the device cannot hold it the host sorter runs. */
int nm_index_build_device(const char *fasta_path, const char *index_path, uint8_t sa_ratio, uint8_t seed_len, int device);

/* Read an index file and upload it to HBM of `device` (>= 0).  seed_len_override: -1 keeps the
 * length recorded in the file (the reference's --seed-length, default 12), -2 = automatic
 * (ceil(log4 n) + 2 bases, at most 16 and at most a quarter of the free HBM; automatic mode also
 * builds the LF blocks and the two quad tables that the sites read), -3 = automatic
 * with small tables (seed <= 15, quad cores <= 13, 20 GB at most: what a one-shot run wants, the large
 * ones can cost seconds of allocation), 0 disables the seed table, 1..16
 * forces a length.  Searches whose shortest length is below the table's get a second small
 * table of that length on first use.  Replaces createIndex()
 * (src/newmap-count.c:9-17) -- but returns an error instead of continuing with a bad handle. */
int nm_index_open(const char *index_path, int device, int seed_len_override, nm_index **out);
/* The same with a budget: the automatic table sizes (seed_len_override -2 / -3) treat `hbm_budget_bytes` as all the HBM there is
 * for this index (file contents included), so that the 2 - 4 index files of one search (newmap/search.py:656-697) co-reside,
 * each with tables for its share.  0 = what the device reports free at the time of the call (nm_index_open). */
int nm_index_open_budget(const char *index_path, int device, int seed_len_override, uint64_t hbm_budget_bytes, nm_index **out);
void nm_index_close(nm_index *ix);

/* index facts: 0 n (BWT length), 1 forward text length, 2 separators, 3 records, 4 raw bases,
 * 5 seed length in use, 6 device bytes held, 7 sa_ratio recorded, 8 range kernel used by the last
 * launch (see NM_OPT_KERNEL), 9 and 12 always 0 (structures of earlier versions), 10 device index,
 * 11 LF blocks in use, 13 repeat probes enabled; of the last
 * range-mode launch's repeat probes (waits for the device; 14..16 need
 * NM_OPT_COUNT_STEPS): 14 LF steps, 15 rank blocks read, 16 seed entries read, 17 positions
 * settled without a search; 18 core length of the quad table (0 = none), 19 core length of the second quad table with
 * short cores (0 = none), 20 core length of the table the sites of the last launch read; 21 fingerprint of the last
 * host-buffer segment call, 22 initial search length, 23 segments that went through the exact guard, 24 / 25 string length
 * and entries of the repeat dictionary (0 = none), 26 two-base LF blocks built (1) or not */
uint64_t nm_index_info(const nm_index *ix, int what);

/* ------------------------------------------------------------------------- compat seam ------
 * Forward-strand occurrence counts, exactly the two functions of src/newmap-count.c.
 * k-mers may contain any byte: a k-mer with a non-ACGT byte counts 0. */
int nm_count_kmers(nm_index *ix, const uint8_t *kmers, const uint64_t *offsets /* n+1 */,
                   uint64_t n, uint32_t *counts_out);
int nm_count_from_sequence(nm_index *ix, const uint8_t *seq, uint64_t seq_len,
                           const uint64_t *starts, const uint64_t *lens, uint64_t n,
                           uint32_t *counts_out);

/* ----------------------------------------------------------------------- fused hot path -----
 * One segment (bytes seq[0..seq_len), of which the first num_kmers are positions and the rest
 * is lookahead): newmap/search.py:383-548 `binary_search` for range mode, :551-644
 * `linear_search` for list mode.  out has num_kmers elements of elem_bytes (1, 2 or 4) bytes
 * (search.py:204-212).  *n_ambiguous = positions whose own byte is not in ACGTacgt (:403).
 * On NM_E_KMER_NOT_FOUND, *bad_pos is the first position whose k-mer is absent from the index. */
int nm_min_unique_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers,
                          uint32_t kmin, uint32_t kmax, uint32_t initial_len, int use_revcomp,
                          int elem_bytes, void *out, uint64_t *n_ambiguous, uint64_t *bad_pos);
int nm_fixed_k_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers,
                       const uint32_t *ks, uint32_t nk, int use_revcomp, int elem_bytes, void *out,
                       uint64_t *n_ambiguous, uint64_t *bad_pos);

/* Several FASTA files processed in lock-step against several index files -- the reference's
 * bisulfite-style mode (newmap/search.py:251-265, 461, 656-697): the count of a position is summed over
 * every (index, sequence) pair and a position is unique when the total equals n_seqs.  seqs[i] are the
 * n_seqs segments (equal length seq_len; mask, upper bound and geometry come from seqs[0]).  ks / nk /
 * range_mode as in nm_search_fasta.  At most 4 indexes (all open on one device) and 4 sequences. */
int nm_search_segment_multi(nm_index *const *indexes, uint32_t n_indexes, const uint8_t *const *seqs,
                            uint32_t n_seqs, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks,
                            uint32_t nk, int range_mode, int use_revcomp, int elem_bytes, void *out,
                            uint64_t *n_ambiguous, uint64_t *bad_pos);

/* newmap/search.py:744-766 + :769-882 in one launch: out[p] (uint32) = per-position inclusive
 * upper search length; ambiguous positions keep kmax. */
int nm_upper_bound_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers,
                           uint32_t kmax, uint32_t *out);

/* --------------------------------------------------------------- device-resident variants ---
 * Same operations with seq/out already in HBM (pointers from hipMalloc or a torch tensor's
 * data_ptr on the SAME device) and asynchronous on `stream` (hipStream_t, NULL = the handle's
 * own stream).  `d_status` points to NM_STATUS_WORDS uint64 in device memory, reset by the
 * call: [0] ambiguous positions, [1] 1 if some k-mer was absent, [2] first absent position
 * (UINT64_MAX if none); and, only after nm_set_option(ix, NM_OPT_COUNT_STEPS, 1) -- the
 * "counter build" of the kernel used for roofline accounting -- [3] LF steps executed,
 * [4] distinct rank structures those steps read (lo and hi in one block count once),
 * [5] 8-byte table words read by the dominant kernel (k_sites: 4 per quad entry; k_min_unique / k_fixed_k: seed entries),
 * [6] strand-block reads (--norc, compat counts) or, after k_sites, the table words k_resolve read (4 per second-chance
 * entry, 1 per seed entry), [7] positions searched.
 * Streams: a call is ordered on the stream it is given and nowhere else.  The handle keeps its launch scratch once per
 * caller stream (up to 5; further streams take over the least recently used set behind an event), so calls given
 * different streams -- independent segments with their own d_out / d_status -- may overlap on the device.  Host-side the
 * calls on one handle are still made one at a time (nm_set_option / info likewise). */
/* A caller that destroys a stream it has passed to the calls below first hands the stream's launch scratch back
 * (waits for the stream, frees the lane for the next stream). */
int nm_stream_release(nm_index *ix, void *stream);
#define NM_STATUS_WORDS 16
#define NM_STATUS_HASH 8           /* [8]: fingerprint of the segment's num_kmers positions (csrc/nm_hash.h); the
                                      driver joins the segments of a record (nm_hash_join) and asks nm_index_has_record */
int nm_min_unique_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers,
                              uint32_t kmin, uint32_t kmax, int use_revcomp, int elem_bytes,
                              void *d_out, uint64_t *d_status, void *stream);
int nm_fixed_k_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers,
                           const uint32_t *ks, uint32_t nk, int use_revcomp, int elem_bytes,
                           void *d_out, uint64_t *d_status, void *stream);

enum {
    NM_OPT_COUNT_STEPS = 1,
    NM_OPT_TIMING = 3,
    NM_OPT_KERNEL = 4,             /* range-mode kernel: 0 automatic (default), 1 one lane per position (k_min_unique),
                                      5 the sites (k_sites + k_resolve: one 128-byte quad-table line per group of
                                      kmin - core length + 1 positions; needs core length + 4 <= kmin <= 252) */
    NM_OPT_FORCE_BIG = 6,          /* test hook: the instantiations for indexes beyond 2^31 rows (what a 3 Gbp genome runs) on a small
                                      index; same results */
    NM_OPT_SEED_POLICY = 7,        /* A/B switches, same results: seed-table load 0 default, 1 non-temporal, 2 agent-scope (sc1);
                                      + 0x800 the blocks of k_sites never walk themselves, + 0x1000 no repeat dictionary, + 0x2000 the
                                      dictionary in the second chance's place.  (The two switches that cut work out of the kernels
                                      and give wrong results, 0x100 / 0x200, exist in the measurement build only -- `make measure`,
                                      libnewmap_amd_measure.so, loaded by tools/ -- this library rejects them.) */
    NM_OPT_LF_BLOCKS = 9,          /* LF steps on the 16-byte LF entries (default when built) or the packed rank blocks */
    NM_OPT_REPEAT_PROBES = 10,     /* both-strand range mode: one probe per 64 positions settles stretches that occur
                                      twice over more than kmax bases (default 1; 0 = every position searches for itself) */
    NM_OPT_LIST_VIA_RANGE = 11,    /* list mode on both strands runs on the range kernels (one length: kmin = kmax = k;
                                      several lengths >= the quad window: the sites + the list form of k_resolve);
                                      default 1, 0 = always the list kernel, for A/B */
    NM_OPT_SITE_TABLE = 13,        /* measurement / tests: which quad table the sites read: 0 pick per launch (default), 1 the one
                                      with long cores, 2 the one with short cores (the other backs it up in k_resolve) */
    NM_OPT_SWEEP = 17,             /* the positions the sites leave open: 1 (default) = launches list the words that hold them and k_sweep takes
                                      the launch where they are dense (neighbours share their walks), once the handle has met open positions;
                                      2 = from the first launch on (tests); 0 = every position walks for itself (k_resolve), for A/B */
    NM_OPT_LCP = 18,               /* A/B: the sweep reads the index's LCP bytes where the end of a chain moves (default 1 where the index file
                                      holds them and the handle is a resident one) or walks */
    NM_OPT_LF2 = 16,               /* A/B: walks take two bases per step on the two-base LF blocks (default 1 where they were built) */
    NM_OPT_SEGMENT_GUARD = 15,     /* default 1: nm_min_unique_segment / nm_fixed_k_segment (host buffers: the seam of binary_search /
                                      linear_search) run the exact zero-count guard unless the segment is a whole indexed record;
                                      0: the caller checks whole records itself (what the drivers do) */
    NM_OPT_INITIAL_LENGTH = 14,    /* --initial-search-length of the run (0 = none): the native drivers hand it to the exact guard, whose
                                      replay of the reference's probe schedule depends on it; results never do */
    NM_OPT_SITE_D = 12             /* measurement / tests: cap (0..59, default 59) on d = kmin - (core length + 4); a site
                                      settles a group of d + 5 positions */
};
int nm_set_option(nm_index *ix, int option, int64_t value);

/* NM_OPT_TIMING = 1 records HIP events on the launch stream, six kinds of start/stop pairs per segment:
 * kind 0 around the search kernel alone (k_sites / k_min_unique / k_fixed_k), kind 1 around ALL kernels of the
 * segment (encode pass, sites, repeat probes, finishing stage), kinds 2 / 3 / 4 around the coarse repeat probes, the fine
 * repeat probes and the finishing stage (k_open_words + k_sweep + k_resolve), kind 5 around k_sweep alone, each on the
 * stream the kernel is launched on (bench.py quotes the roofline figure for the kernel with the largest total).
 * nm_timing_read_kind waits for the events of one kind, returns their number, summed and longest duration in
 * ms, and resets that record; nm_timing_read = kind 0. */
int nm_timing_read(nm_index *ix, uint64_t *n_launches, double *total_ms, double *max_ms);
int nm_timing_read_kind(nm_index *ix, int kind, uint64_t *n_launches, double *total_ms, double *max_ms);

/* --------------------------------------------------------------- native search driver --------
 * The whole of newmap/search.py:197-380 `write_unique_counts` for one FASTA and one index in one
 * call (record and segment rules of newmap/fasta.py:20-190), one `<out_dir>/<id>.unique.uint8|16|32` per
 * record id.  A plain FASTA file is mapped and stripped by a pool of host threads, its segments go through a
 * ring of pinned slots (H2D / kernels / D2H on one stream) and a pool of writer threads puts the results into
 * the files with pwrite; gzip input takes a streaming reader (one thread, two slots).  `ks`: the range kmin..kmax is
 * given by its two ends when range_mode != 0, else the list of lengths in order.  include / exclude:
 * record ids to keep / to skip (at most one of the two lists non-empty).  `cb` (may be NULL) is
 * called once per output file, in file order, after the search; `total` receives the sums. */
typedef struct nm_search_summary {
    uint64_t records, positions, ambiguous, unique, no_unique;
    uint32_t max_len, min_len;      /* search.py:242-243: start at kmin / kmax, updated by found lengths */
} nm_search_summary;
typedef void (*nm_record_callback)(const char *record_id, const nm_search_summary *s, void *user);
int nm_search_fasta(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks,
                    uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                    const char *const *include_ids, uint32_t n_include,
                    const char *const *exclude_ids, uint32_t n_exclude,
                    nm_record_callback cb, void *user, nm_search_summary *total);

/* The same call as ONE RANK of `world` (one process per GPU): the position space of the selected records is cut into
 * interleaved chunks of ~64 M positions, rank r owns chunks r, r + world, ...; it strips only the byte ranges of its
 * own work units out of the (uncompressed) FASTA, searches them on its GPU and writes them at their offsets into the
 * per-record files, which every rank opens and sizes alike.  No collective; the ranks need a common file system.
 * `cb` / `total` cover this rank's share.  Replaces nothing in the reference (it has no multi-device mode):
 * SURVEY.md section 8(e). */
int nm_search_fasta_shard(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks,
                          uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                          const char *const *include_ids, uint32_t n_include,
                          const char *const *exclude_ids, uint32_t n_exclude,
                          nm_record_callback cb, void *user, nm_search_summary *total, int rank, int world);

/* --------------------------------------------------------------- track on the device ----------
 * newmap/track.py:22-121 for ONE unique-length file: marks, windowed prefix sums, run-length BED
 * lines and formatted WIG lines, appended to bed_path / wig_path (either may be NULL).  `chr_name` is
 * the text before ".unique" in the file name (track.py:155-158). */
int nm_track_file(int device, const char *unique_path, const char *chr_name, int elem_bytes, uint32_t k,
                  const char *bed_path, const char *wig_path, uint64_t *n_positions, uint64_t *n_runs);

/* small device-memory helpers so a host program needs no other HIP binding */
int nm_dev_alloc(int device, uint64_t bytes, void **out);
int nm_dev_free(int device, void *p);
int nm_dev_upload(int device, void *dst, const void *src, uint64_t bytes);
int nm_dev_download(int device, void *dst, const void *src, uint64_t bytes);
int nm_dev_sync(int device);
uint64_t nm_dev_free_bytes(int device);                 /* free HBM right now (0 on error): budgets for nm_index_open_budget */
int nm_device_count(void);

/* ------------------------------------------------------------------ record fingerprints -----
 * Replaces the unconditional zero-count check of newmap/search.py:699-722 by "is this record one of the indexed
 * records?": an index file (format 2) lists the length and fingerprint of each of its records (csrc/nm_hash.h: a
 * sum over the record's 64-base words of mix(planes) * R^word mod 2^64).  Every segment call leaves the fingerprint of
 * its positions in d_status[NM_STATUS_HASH]; the driver joins the segments of a record -- H(a.b) = H(a) + R^(len(a)/64) *
 * H(b) for segments that start at multiples of 64 bases of their record, which is how the native drivers cut them -- and
 * looks the record up.  A record that is found holds no absent k-mer; any other record is searched again with every probe of the
 * reference's schedule verified (nm_guard_segment_dev below). */
int nm_index_has_record(const nm_index *ix, uint64_t length, uint64_t fingerprint);      /* 1 = indexed, 0 = not */
/* The exact guard over one segment: replays, per position, the probe schedule of newmap/search.py:383-548 (range mode:
 * ks = {kmin, kmax}; initial_len = --initial-search-length or 0) or :551-644 (list mode) and walks the longest probed
 * k-mer; NM_E_KMER_NOT_FOUND (host-buffer form) / d_status[1], [2] (device form) exactly when the reference would raise. */
int nm_guard_segment_dev(nm_index *ix, const void *d_seq, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks, uint32_t nk,
                         int range_mode, uint32_t initial_len, int use_revcomp, uint64_t *d_status, void *stream);
int nm_guard_segment(nm_index *ix, const uint8_t *seq, uint64_t seq_len, uint64_t num_kmers, const uint32_t *ks, uint32_t nk,
                     int range_mode, uint32_t initial_len, int use_revcomp, uint64_t *bad_pos);
/* nm_search_fasta does all of this itself.  For jobs of several ranks: nm_search_fasta_shard_ex also hands back, per FASTA
 * record with data (file order), rec_info[3 i + 0 .. 2] = {length, fingerprint summed over THIS rank's units, 1 if
 * searched}; the caller adds the ranks' sums, asks nm_index_has_record, and runs nm_guard_fasta (flags[i] != 0 = guard
 * record i) on one rank. */
int nm_search_fasta_shard_ex(nm_index *ix, const char *fasta_path, const char *out_dir, const uint32_t *ks, uint32_t nk, int range_mode,
                             int use_revcomp, uint64_t batch, const char *const *include_ids, uint32_t n_include,
                             const char *const *exclude_ids, uint32_t n_exclude, nm_record_callback cb, void *user,
                             nm_search_summary *total, int rank, int world, uint64_t *rec_info, uint64_t capacity, uint64_t *n_records);
int nm_guard_fasta(nm_index *ix, const char *fasta_path, const uint32_t *ks, uint32_t nk, int range_mode, int use_revcomp, uint64_t batch,
                   const char *const *include_ids, uint32_t n_include, const char *const *exclude_ids, uint32_t n_exclude,
                   const uint8_t *flags, uint64_t n_flags);
uint64_t nm_index_records(const nm_index *ix, uint64_t *lengths, uint64_t *fingerprints, uint64_t capacity);   /* returns their number */
uint64_t nm_fingerprint_join(uint64_t fp_a, uint64_t len_a, uint64_t fp_b);
uint64_t nm_fingerprint_sequence(const uint8_t *seq, uint64_t len);                      /* host loop, for small inputs and tests */

#ifdef __cplusplus
}
#endif
#endif /* NEWMAP_AMD_H */

// nm_format.h -- on-disk / in-HBM layout of the both-strand FM-index (shared by the host
// builder nm_build.cpp and the device engine nm_engine.hip).
//
// Text model (what the reference's index means; pinned by
// reference tests/test_count_kmers.py:21-25): the FASTA is a set of records; a k-mer occurs only
// inside one record, case-insensitively.  Every maximal ACGT run r_1..r_k of the records (record
// boundaries and non-ACGT runs both end a run) goes into
//
//      T  =  r_1 $ r_2 $ ... r_k $   rc(r_k) $ ... rc(r_1) $   #
//            \------ F, n_fwd -----/ \------- RC, n_fwd -------/
//
// so that ONE backward search of a pattern in T yields count_fwd(P) + count_fwd(revcomp(P)), the
// total newmap/search.py:647-697 obtains with two library calls.  '$' and '#' never occur in a
// pattern; they sort before A.
//
// Rank structure: the BWT of T, 64 positions per 32-byte block:
//      u32 cnt[4]   occurrences of A,C,G,T in BWT[superblock start .. block start)
//                   (bit 31 of cnt[0] = "this block holds a separator")
//      u64 lo, hi   bit-planes of the 2-bit codes of the 64 positions (separator stored as A)
// A superblock is 2^31 positions; absolute counts at superblock starts (plus C[]) live in a
// tiny table.  Separator positions of the BWT are listed, sorted, in `sep`; blocks flagged above
// correct rank(A) from that list.
// Strand structure: 64 suffix-array positions per 16-byte block: u64 count of RC-half suffixes
// before the block, u64 bits (1 = suffix starts in the RC half).  Used by forward-only counts
// (count_kmers*, --norc).
// LCP bytes (optional section behind the record list, off_lcp): per row the number of bases its suffix shares with the suffix
// of the row before it, capped at 255.  A row that holds the ONLY occurrence of a string tells with its two bytes how
// long that string has to be to occur once (nm_core.h "the sweep").
#ifndef NM_FORMAT_H
#define NM_FORMAT_H

#include <stdint.h>

#define NM_MAGIC "NMAPGFX1"
#define NM_FORMAT_VERSION 2u        /* 2: + the record list (length, fingerprint) behind the separators */
#define NM_SUPER_SHIFT 31
#define NM_MAX_SUPER 16            /* up to 2^35 BWT positions */
#define NM_SEP_FLAG 0x80000000u
#define NM_LCP_CAP 255u

struct nm_rank_block {             /* 32 bytes, 64 BWT positions */
    uint32_t cnt[4];
    uint64_t lo, hi;
};

/* LF block, built on the device when the index is opened: per 64 BWT rows and per base c one
 * 16-byte entry {C[c] + occurrences of c before the block, indicator bits of c}.  LF_c(i) is then ONE
 * 16-byte load, one popcount and one add -- half the load instructions of the packed rank block
 * (walks with wide intervals are bound by the L1's divergent-address rate), no
 * superblock table, and separators need no exception path (their rows match no base).
 * Layout: entry (64-row block b, base c) at index b * 4 + c. */
struct nm_lf_entry {
    uint64_t base;
    uint64_t bits;
};

struct nm_strand_block {           /* 16 bytes, 64 suffix-array positions */
    uint64_t before;
    uint64_t bits;
};

struct nm_file_header {            /* 1024 bytes */
    char     magic[8];
    uint32_t version;
    uint32_t header_bytes;
    uint64_t n;                    /* BWT length = 2*n_fwd + 1 */
    uint64_t n_fwd;                /* suffixes at text positions [n_fwd, 2*n_fwd) are RC-half */
    uint64_t n_sep;                /* separators in T (including '#') */
    uint64_t base_count[4];        /* A,C,G,T in T */
    uint64_t n_rank_blocks;
    uint64_t n_strand_blocks;
    uint64_t n_records;            /* FASTA records with data */
    uint64_t raw_bases;            /* sum of record lengths as in the FASTA */
    uint64_t n_runs;               /* k */
    uint64_t off_rank, off_strand, off_sep, file_bytes;
    uint8_t  sa_ratio, seed_len, pad8[6];
    uint64_t n_super;
    uint64_t super_cnt[NM_MAX_SUPER][4];   /* A,C,G,T before each superblock */
    uint64_t off_records;          /* n_records entries of nm_record_entry (format 2, nm_hash.h) */
    uint64_t off_lcp;              /* 0 = none; else n + 1 bytes: lcp[j] = bases the suffixes of rows j - 1 and j share, capped at
                                      NM_LCP_CAP (lcp[0] = lcp[n] = 0): what the sweep reads when the end of a chain moves */
    uint8_t  reserved[1024 - 8 - 8 - 8 * 3 - 32 - 8 * 5 - 8 * 4 - 8 - 8 - NM_MAX_SUPER * 32 - 8 - 8];
};

struct nm_record_entry {           /* one FASTA record with data: its length in bytes and the fingerprint of nm_hash.h */
    uint64_t length;
    uint64_t hash;
};

#ifdef __cplusplus
static_assert(sizeof(nm_rank_block) == 32, "rank block must be 32 bytes");
static_assert(sizeof(nm_strand_block) == 16, "strand block must be 16 bytes");
static_assert(sizeof(nm_lf_entry) == 16, "LF entry must be 16 bytes");
static_assert(sizeof(nm_file_header) == 1024, "header must be 1024 bytes");
#endif

#endif

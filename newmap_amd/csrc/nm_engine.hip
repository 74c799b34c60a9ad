// nm_engine.hip -- the MI355X (gfx950 / CDNA4) engine behind the C-ABI of include/newmap_amd.h.
//
// Kernels (all integer / bit work, HBM-gather bound, no MFMA):
//   k_encode16 / k_encode   sequence bytes -> 2 bit-planes + ambiguity plane, 32 B / 64 bases
//   k_sites                 range mode: ONE quad-table entry per group of kmin - m + 1 positions settles the group
//                           (nm_core.h "sites"); writes the elements and the bitmap of positions left open
//   k_period_runs           tandem arrays: the period is read off the encoded words, ONE walk per run of strides settles it
//   k_repeat_probe(_coarse) one walk per 64 (512) positions where the bitmap is dense: settles long repeats,
//                           fixes the lengths between equal ends
//   k_open_words            sorts the words of the need bitmap: dense ones for the sweep (longest chains first), sparse ones for the walks
//   k_sweep                 open positions right to left, neighbours sharing their walks: one step to the left on the rows the right
//                           neighbour left; where the end moves, the index's LCP bytes (nm_core.h "the sweep")
//   k_resolve               the positions k_sites left open (all of them, or the sparse words): probe words, else seed table + walk
//                           (the six together: newmap/search.py:383-548)
//   k_min_unique            range mode, one lane per genome position (--norc, kmin below the table's window, A/B)
//   k_fixed_k               list mode,  one lane per genome position (newmap/search.py:551-644)
//   k_guard                 the exact zero-count check of newmap/search.py:699-722 for records that are not indexed ones
//   k_segment_hash          record fingerprints (nm_hash.h) on the paths that do not run k_sites
//   k_multi                 several FASTA files x several index files (newmap/search.py:461, 656-697)
//   k_count                 forward-strand counts of (start, len) k-mers (src/newmap-count.c:91-206)
//   k_upper                 per-position upper search length (newmap/search.py:744-882)
//   k_seed, k_seed_level, k_quad_build, k_lf_blocks   tables built at open
// The per-position logic lives in nm_core.h.
#include <hip/hip_runtime.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <new>
#include <thread>
#include <vector>
#include <unistd.h>

#include "../../include/newmap_amd.h"
#include "nm_format.h"
#include "nm_internal.h"

extern "C" int nm_index_has_record(const nm_index *ix, uint64_t length, uint64_t hash);
#define NM_HD __device__ __forceinline__
#define NM_HASH_FN __host__ __device__ __forceinline__      /* the fingerprint helpers also run on the host (tables, joins) */
struct nm_view;
static __device__ __forceinline__ uint64_t nm_seed_load_policy(const nm_view &ix, uint64_t slot);
#define NM_SEED_LOAD(ix, slot) nm_seed_load_policy((ix), (slot))
// the two 32-byte reads of a turn of the sweep (nm_core.h: nm_sweep_step): issued back to back, ONE wait
typedef unsigned int nm_u32x4 __attribute__((ext_vector_type(4)));
struct nm_q4_raw { nm_u32x4 lo, hi; };
#define NM_Q4_ZERO(r) ((r).lo = (r).hi = nm_u32x4{0u, 0u, 0u, 0u})
#define NM_Q4_ISSUE(r, p) asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"((r).lo), "=&v"((r).hi) : "v"(p) : "memory")
#define NM_Q2_ISSUE(r, p) do { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"((r).lo) : "v"(p) : "memory"); (r).hi = nm_u32x4{0u, 0u, 0u, 0u}; } while (0)
#define NM_Q4_WAIT2(a, b) asm volatile("s_waitcnt vmcnt(0)" : "+v"((a).lo), "+v"((a).hi), "+v"((b).lo), "+v"((b).hi) : : "memory")
#define NM_Q4_VALUE(r) nm_q4_of_raw(r)
struct nm_q4;
static __device__ __forceinline__ nm_q4 nm_q4_of_raw(const nm_q4_raw &r);
#include "nm_core.h"
static __device__ __forceinline__ nm_q4 nm_q4_of_raw(const nm_q4_raw &r) {
    nm_q4 q;
    q.x[0] = (uint64_t)r.lo.x | ((uint64_t)r.lo.y << 32); q.x[1] = (uint64_t)r.lo.z | ((uint64_t)r.lo.w << 32);
    q.x[2] = (uint64_t)r.hi.x | ((uint64_t)r.hi.y << 32); q.x[3] = (uint64_t)r.hi.z | ((uint64_t)r.hi.w << 32);
    return q;
}

// seed-table gather under a selectable cache policy (NM_OPT_SEED_POLICY; measurement knob)
static __device__ __forceinline__ uint64_t nm_seed_load_policy(const nm_view &ix, uint64_t slot) {
    const uint64_t *p = ix.seed + slot;
    if (ix.seed_policy == 1) return __builtin_nontemporal_load(p);
    if (ix.seed_policy == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// One translation unit, five parts (the kernels are templates launched from the host code next to them):
#include "nm_kernels.hip.h"     // device helpers + all kernels
#include "nm_handle.hip.h"      // nm_index, lanes, timing events
#include "nm_tables.hip.h"      // seed / quad / LF tables, open, close, info, options
#include "nm_launch.hip.h"      // launch order of a segment: sites, probes, resolve, fingerprints
#include "nm_abi.hip.h"         // segment / count / guard entry points

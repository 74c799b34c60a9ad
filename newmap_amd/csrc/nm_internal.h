// nm_internal.h -- shared between the host builder and the device engine
#ifndef NM_INTERNAL_H
#define NM_INTERNAL_H
#ifdef __cplusplus
extern "C++" {
#endif
#include <stdint.h>
void nm_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
// suffix array of T[0..n) (symbols 0..5, T[n-1] == 0 unique) supplied by another translation unit
typedef int (*nm_sa32_provider)(const uint8_t *T, uint64_t n, int32_t *SA, void *ctx);
// for texts of 2^31 symbols or more: the provider fills bw[j] = T[SA[j] - 1] | (SA[j] in [n_fwd, 2 n_fwd) ? 0x80 : 0),
// one byte per suffix-array position -- what the block construction consumes -- so that the suffix array itself
// (8 bytes per symbol) never leaves the device.  A provider that cannot take the text returns NM_E_ALLOC /
// NM_E_TOO_LARGE and the host sorter takes over.
typedef int (*nm_bwt_provider)(const uint8_t *T, uint64_t n, uint64_t n_fwd, uint8_t *bw, void *ctx);
// LCP bytes of the index being built (nm_format.h: off_lcp): a provider that has the suffix array on the device fills the host
// buffer (n + 1 bytes, nullptr = not wanted) and says so; otherwise the builder computes them from its own suffix array
uint8_t *nm_build_lcp_buffer(void);
void nm_build_lcp_done(void);
int nm_index_build_impl(const char *fasta_path, const char *index_path, uint8_t sa_ratio, uint8_t seed_len,
                        nm_sa32_provider provider, void *provider_ctx, nm_bwt_provider big_provider = nullptr);
#ifdef __cplusplus
}
#endif
#endif

// nm_internal.h -- shared between the host builder and the device engine
#ifndef NM_INTERNAL_H
#define NM_INTERNAL_H
#ifdef __cplusplus
extern "C++" {
#endif
void nm_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#ifdef __cplusplus
}
#endif
#endif

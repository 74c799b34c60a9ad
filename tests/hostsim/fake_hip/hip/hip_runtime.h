// tests/hostsim/fake_hip/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
// A stand-in for the HIP runtime that lets the HOST logic of newmap_amd/csrc/nm_driver.hip (FASTA scan, units, worker
// threads, pinned slots, hand-over to the "device", pwrite of the results, record fingerprints, guard pass) be compiled
// with g++ and run under ThreadSanitizer / AddressSanitizer -- GPU sanitizers do not exist on the pool.  Memory is host
// memory, copies are memcpy, streams and events are empty (every "asynchronous" call completes before it returns), the one
// kernel of the driver (k_out_summary) is replaced by a host loop inside nm_driver.hip (NM_DRIVER_HOST_SUMMARY).  The engine
// entry points the driver calls are stubbed in tests/hostsim/driver_sim.cpp.
#ifndef NM_FAKE_HIP_RUNTIME_H
#define NM_FAKE_HIP_RUNTIME_H
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0 };
typedef struct fake_stream *hipStream_t;
typedef struct fake_event *hipEvent_t;
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
enum { hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipEventBlockingSync = 1 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
#define __global__
#define __restrict__

static inline const char *hipGetErrorString(hipError_t) { return "fake HIP error"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
#endif

// Host stand-in for the slice of the HIP device dialect that add-gym_amd/csrc/rigid.hip uses, so that the SAME source text runs on the
// CPU under AddressSanitizer / UndefinedBehaviorSanitizer / MemorySanitizer (tools/slp_repro/README.md).  A workgroup is 64 host
// threads; __syncthreads is a 64-thread barrier; the quad shuffles exchange through a per-quad mailbox between two quad barriers
// (every lane of a quad reaches every shuffle: the kernel's shuffle sites are wave-uniform).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <pthread.h>

#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(x)
#define __shared__
#define __restrict__ __restrict

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipDeviceAttributeMultiprocessorCount = 0, hipFuncAttributeMaxDynamicSharedMemorySize = 0 };
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "host"; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return 0; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
#define hipLaunchKernelGGL(...) ((void)0)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, o, s) (*(p))
#define __hip_atomic_store(p, v, o, s) (*(p) = (v))
inline void __threadfence() {}
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }

namespace hostsim {
struct Tid { unsigned x, y, z; };
extern thread_local Tid tid, bid, bdim, gdim;
extern pthread_barrier_t wg_barrier;        // 64 lanes
extern pthread_barrier_t quad_barrier[16];  // 4 lanes each
extern float mailbox[64];
extern unsigned mailbox_u[64];
}  // namespace hostsim
#define threadIdx hostsim::tid
#define blockIdx hostsim::bid
#define blockDim hostsim::bdim
#define gridDim hostsim::gdim
// (`extern __shared__ float lds[];` inside a kernel binds to the array host_run.cpp defines in the same unnamed namespace)
inline void __syncthreads() { pthread_barrier_wait(&hostsim::wg_barrier); }
inline float __shfl(float v, int src, int) {
  const int me = hostsim::tid.x;
  hostsim::mailbox[me] = v;
  pthread_barrier_wait(&hostsim::quad_barrier[me >> 2]);
  const float r = hostsim::mailbox[src];
  pthread_barrier_wait(&hostsim::quad_barrier[me >> 2]);
  return r;
}
inline float __shfl_xor(float v, int m, int) { return __shfl(v, (int)hostsim::tid.x ^ m, 64); }
inline unsigned __shfl_xor(unsigned v, int m, int) {
  const int me = hostsim::tid.x;
  hostsim::mailbox_u[me] = v;
  pthread_barrier_wait(&hostsim::quad_barrier[me >> 2]);
  const unsigned r = hostsim::mailbox_u[me ^ m];
  pthread_barrier_wait(&hostsim::quad_barrier[me >> 2]);
  return r;
}
inline int min(int a, int b) { return a < b ? a : b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline int __any(int pred) {  // wave-wide vote (the one-lane kernel): all 64 lanes reach it
  const int me = hostsim::tid.x;
  hostsim::mailbox_u[me] = pred != 0;
  pthread_barrier_wait(&hostsim::wg_barrier);
  unsigned r = 0;
  for (int i = 0; i < 64; ++i) r |= hostsim::mailbox_u[i];
  pthread_barrier_wait(&hostsim::wg_barrier);
  return (int)r;
}

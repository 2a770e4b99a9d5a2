// Counter-based random numbers shared by the fill kernels (learn.hip) and the rigid engine's domain randomisation (rigid.hip):
// element q of stream `stream_id` under `seed` is a pure function of (q, stream_id, seed), so a replayed hipGraph, a call-by-call
// run and any launch geometry draw the same numbers.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace {
// Philox4x32-10
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox(uint64_t ctr, uint64_t stream_id, uint64_t seed) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = (uint32_t)stream_id, c3 = (uint32_t)(stream_id >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)

}  // namespace

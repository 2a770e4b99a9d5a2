// Microbenchmark: sustained v_mfma_f32_32x32x16_bf16 rate on gfx950 with operands in registers (what the chip's power
// management leaves of the 2.5 PFLOP/s nominal peak), 1 or 2 waves per SIMD, zero or random data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void k(float* out, const uint4* in, int iters) {
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;
  bf16x8 fa[2][3], fb[2][3];
  for (int a = 0; a < 2; ++a) for (int p = 0; p < 3; ++p) {
    fa[a][p] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * (a * 3 + p)) & 4095]);
    fb[a][p] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * (a * 3 + p) + 777) & 4095]);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
      }
  }
  float s = 0;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int x = 0; x < 16; ++x) s += acc[a][b][x];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out; uint4* in;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 4096 * 16);
  for (int rnd = 0; rnd < 2; ++rnd) {
    std::vector<unsigned short> h(4096 * 8);
    for (auto& v : h) v = rnd ? (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15)) : 0;  // bf16 in [0.0078, 0.03)
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
      const int iters = 4000, grid = 256 * blocks_per_cu;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, in, 100);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, in, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flop = (double)grid * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
      printf("%s data, %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s bf16 = %.1f TFLOP/s of fp32 products (/6)\n", rnd ? "random" : "zero", blocks_per_cu, ms,
             flop / ms / 1e9, flop / ms / 1e9 / 6);
    }
  }
  return 0;
}

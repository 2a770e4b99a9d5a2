// Microbenchmark: what the chip sustains (power management included) on the six-product pattern of the exact bf16 split, operands in
// registers, for (a) the product order, (b) the MFMA shape 32x32x16 vs 16x16x32, (c) the operand data (zeros, random bits, plane-like).
// hipcc --offload-arch=gfx950 -O3 mfma_x3_variants.hip -o mfma_x3_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// V = 0: 32x32x16, order (1,1)(0,2)(2,0)(0,1)(1,0)(0,0)   V = 1: 32x32x16, every consecutive pair shares an operand
// V = 2: 16x16x32 (same tile 64x64 per wave = 4x4 accumulators of 16x16, k 32 per instruction), order as V = 0
// V = 3: f16 32x32x16, FOUR products (hi*hi, hi*lo, lo*hi, lo*lo) of a two-way fp16 split
template <int V>
__global__ __launch_bounds__(256) void k(float* out, const uint4* in, int iters) {
  float s = 0;
  if constexpr (V == 0 || V == 1) {
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
#define M_(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][i], fb[b][j], acc[a][b], 0, 0, 0)
          if (V == 0) { M_(1, 1); M_(0, 2); M_(2, 0); M_(0, 1); M_(1, 0); M_(0, 0); }
          else { M_(0, 2); M_(0, 1); M_(1, 1); M_(1, 0); M_(2, 0); M_(0, 0); }
#undef M_
        }
    }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int x = 0; x < 16; ++x) s += acc[a][b][x];
  } else if constexpr (V == 2) {
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int x = 0; x < 4; ++x) acc[a][b][x] = 0.f;
    bf16x8 fa[4][3], fb[4][3];
    for (int a = 0; a < 4; ++a) for (int p = 0; p < 3; ++p) {
      fa[a][p] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * (a * 3 + p)) & 4095]);
      fb[a][p] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * (a * 3 + p) + 777) & 4095]);
    }
    for (int it = 0; it < iters; it += 2) {  // one pass = k 32: half as many passes for the same arithmetic
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#define M_(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a][i], fb[b][j], acc[a][b], 0, 0, 0)
          M_(1, 1); M_(0, 2); M_(2, 0); M_(0, 1); M_(1, 0); M_(0, 0);
#undef M_
        }
    }
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int x = 0; x < 4; ++x) s += acc[a][b][x];
  } else {
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;
    f16x8 fa[2][2], fb[2][2];
    for (int a = 0; a < 2; ++a) for (int p = 0; p < 2; ++p) {
      fa[a][p] = __builtin_bit_cast(f16x8, in[(threadIdx.x + 64 * (a * 3 + p)) & 4095]);
      fb[a][p] = __builtin_bit_cast(f16x8, in[(threadIdx.x + 64 * (a * 3 + p) + 777) & 4095]);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#define M_(i, j) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a][i], fb[b][j], acc[a][b], 0, 0, 0)
          M_(1, 1); M_(0, 1); M_(1, 0); M_(0, 0);
#undef M_
        }
    }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int x = 0; x < 16; ++x) s += acc[a][b][x];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V> void run(float* out, uint4* in, const char* what, const char* data, int products) {
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int iters = 4000, grid = 256 * bpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, in, 400);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // fp32-product work: a 64x64 tile per wave, k 16 per iteration
    const double fl32 = (double)grid * 4 * iters * 2.0 * 64 * 64 * 16;
    printf("%-34s %-12s %d wave/SIMD: %7.3f ms  %7.1f TFLOP/s MFMA = %6.1f TFLOP/s of fp32 products\n", what, data, bpc, ms, fl32 * products / ms / 1e9, fl32 / ms / 1e9);
  }
}

int main() {
  float* out; uint4* in;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 4096 * 16);
  const char* names[4] = {"zeros", "narrow", "random-bits", "half-zero"};
  for (int d = 0; d < 4; ++d) {
    std::vector<unsigned short> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) {
      unsigned short v = 0;
      if (d == 1) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));          // bf16 in [0.0078, 0.03), either sign
      if (d == 2) v = (unsigned short)((rand() & 0x7fff) % 0x4700 + 0x0800 + ((rand() & 1) << 15)); // exponents all over (finite in bf16 and fp16)
      if (d == 3) v = ((i / 8) & 1) ? 0 : (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));  // every other 8-chunk zero (ReLU-like)
      h[i] = v;
    }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<0>(out, in, "bf16 32x32x16 x6, ascending order", names[d], 6);
    run<1>(out, in, "bf16 32x32x16 x6, shared operands", names[d], 6);
    run<2>(out, in, "bf16 16x16x32 x6", names[d], 6);
    run<3>(out, in, "f16  32x32x16 x4 (two-way split)", names[d], 4);
  }
  return 0;
}

// Microbenchmark: what limits the fp32 MFMA GEMM main loop on gfx950?  Variants add one ingredient at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 128 * 36];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;
  for (int i = threadIdx.x; i < 2 * 2 * 128 * 36; i += 256) lds[i] = in[(blockIdx.x * 977 + i) & 0xfffff];
  __syncthreads();
  float4 fa[2], fb[2];
  fa[0] = fa[1] = fb[0] = fb[1] = make_float4(in[lane], in[lane + 64], in[lane + 128], in[lane + 192]);
  float4 g0 = make_float4(0, 0, 0, 0);
  const float4* gin = reinterpret_cast<const float4*>(in) + (size_t)blockIdx.x * 4096 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const float* a_cur = lds + (it & 1) * 2 * 128 * 36;
    const float* b_cur = a_cur + 128 * 36;
    float4 r[8];
    if (V >= 3) {
#pragma unroll
      for (int q = 0; q < 8; ++q) r[q] = gin[(size_t)(it & 7) * 2048 + q * 256];
    }
#pragma unroll
    for (int kk = 0; kk < 32; kk += 8) {
      if (V >= 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = *reinterpret_cast<const float4*>(a_cur + ((wave >> 1) * 64 + a * 32 + li) * 36 + kk + 4 * lh);
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[b] = *reinterpret_cast<const float4*>(b_cur + ((wave & 1) * 64 + b * 32 + li) * 36 + kk + 4 * lh);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
        }
    }
    if (V >= 3) {
      float* w = lds + ((it + 1) & 1) * 2 * 128 * 36;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        int f = threadIdx.x + 256 * q;
        *reinterpret_cast<float4*>(w + (f >> 3) * 36 + (f & 7) * 4) = r[q];
      }
    }
    if (V >= 2) __syncthreads();
  }
  float s = g0.x;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int x = 0; x < 16; ++x) s += acc[a][b][x];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(const char* name, int blocks, int iters, float* out, const float* in) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double flop = (double)blocks * 4 * iters * 64 * 4096.0;
  printf("%-40s blocks=%d iters=%d  %8.1f us  %7.1f TFLOP/s\n", name, blocks, iters, ms * 1e3, flop / ms / 1e9);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 4096 * 256 * 4);
  hipMalloc(&in, (size_t)4096 * 4096 * 16 + (1 << 24));
  { size_t n = ((size_t)4096 * 4096 * 16 + (1 << 24)) / 4; float* h = (float*)malloc(n * 4); unsigned x = 12345; bool zero = getenv("ZERO") != nullptr; for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = zero ? 0.f : ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)); } hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice); free(h); }
  for (int blocks : {512, 1024}) {
    run<0>("mfma only (regs)", blocks, 32, out, in);
    run<1>("+ LDS fragment reads", blocks, 32, out, in);
    run<2>("+ barrier per tile", blocks, 32, out, in);
    run<3>("+ global loads + LDS writes", blocks, 32, out, in);
  }
  return 0;
}

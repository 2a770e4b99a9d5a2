// GEMM on bf16 operands STORED as bf16 in HBM (agent.matmul_precision = bf16: bf16 hidden activations, pre-activation gradients
// and weight shadows; fp32 master weights, fp32 accumulation, fp32 or bf16 results):
//   C[M,N] = alpha * sum_k A(m,k) * B(n,k),  A, B bf16, either operand k-contiguous or m/n-contiguous,
// same epilogues, ReLU sign bits, fused bias-gradient column sums and split-K slabs as gemm.hip / gemm_split.hip.
//
// Unlike gemm_split.hip (fp32 operands split into bf16 planes on their way into LDS, 16-deep stages) nothing is converted
// here: 128x128 tile, 4 wavefronts of 64x64 (2x2 accumulators of v_mfma_f32_32x32x16_bf16), 64-deep K stages = 16 MFMAs per
// wave and barrier, LDS double buffer (2 x 2 x 18 KB -> 2 workgroups per CU), next stage prefetched global -> VGPR.
// LDS image of an operand tile: [row][64 k] bf16, row stride 144 B (36 dwords: a ds_read_b128 lane group covers all 64 banks).
// k-contiguous operands arrive as 16-byte chunks and go to LDS unchanged; m/n-contiguous operands (the weight-gradient GEMMs:
// dW = dz^T x, both operands row-major over the minibatch) are transposed on the way: each thread loads the same 8 rows of
// two consecutive k and writes 8 packed (k, k+1) dwords.
//
// Replaces (in bf16 mode): the torch.nn.Linear forward/backward of PPOModel / ADDModel (ppo_model.py:13-21, add_model.py:12-15).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int RS = BK * 2 + 16;         // bytes per LDS row
constexpr int TILE = 128 * RS;          // bytes per operand tile
constexpr int EPI_RUNTIME = -1;

__device__ __forceinline__ unsigned pack_lo(unsigned a, unsigned b) { return (a & 0xffffu) | (b << 16); }   // low halves
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); }  // high halves
// fp32 -> bf16, round to nearest even (no NaN special-casing: the callers' values are finite)
__device__ __forceinline__ u16 to_bf16(float v) {
  const unsigned u = __float_as_uint(v);
  return (u16)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// ---- k-contiguous operand P[r*ld + k]: 128 rows x 8 chunks of 8 k = 1024 chunks, 4 per thread
template <bool GUARD>
__device__ __forceinline__ void load_kc(uint4* reg, const u16* __restrict__ P, int ld, int r0, int k0, int R, int kend) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    const int row = c >> 3, kq = (c & 7) * 8;
    const int r = min(r0 + row, R - 1), k = k0 + kq;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!GUARD || k < kend) v = *reinterpret_cast<const uint4*>(P + (size_t)r * ld + k);  // K % 8 == 0: a chunk is in or out as a whole
    reg[i] = v;
  }
}
__device__ __forceinline__ void store_kc(char* lds, const uint4* reg) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = threadIdx.x + 256 * i;
    *reinterpret_cast<uint4*>(lds + (c >> 3) * RS + (c & 7) * 16) = reg[i];
  }
}
// ---- m/n-contiguous operand P[k*ld + r]: 32 k-pairs x 16 chunks of 8 rows = 512 tasks, 2 per thread (2 loads each)
template <bool GUARD>
__device__ __forceinline__ void load_mc(uint4* reg, const u16* __restrict__ P, int ld, int r0, int k0, int R, int kend) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = threadIdx.x + 256 * i;
    const int kp = t & 31, rq = (t >> 5) * 8;
    const int r = min(r0 + rq, R - 8), k = k0 + 2 * kp;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (!GUARD || k + j < kend) v = *reinterpret_cast<const uint4*>(P + (size_t)(k + j) * ld + r);
      reg[2 * i + j] = v;
    }
  }
}
__device__ __forceinline__ void store_mc(char* lds, const uint4* reg) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = threadIdx.x + 256 * i;
    const int kp = t & 31, rq = (t >> 5) * 8;
    const uint4 a = reg[2 * i], b = reg[2 * i + 1];  // rows rq..rq+7 at k and k+1
    const unsigned av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
    char* dst = lds + rq * RS + kp * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<unsigned*>(dst + (2 * q) * RS) = pack_lo(av[q], bv[q]);
      *reinterpret_cast<unsigned*>(dst + (2 * q + 1) * RS) = pack_hi(av[q], bv[q]);
    }
  }
}

__device__ __forceinline__ bf16x8 frag(const char* lds, int row, int kstep, int h) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + row * RS + kstep * 32 + h * 16));
}

template <bool AKC, bool BKC, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(addhip_gemm_t g, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE];
  constexpr int STAGE = 2 * TILE;

  // XCD-aware remap (blocks b and b+8 share an XCD): each XCD gets a contiguous run of tiles, N-tile fastest
  const int total = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q = total >> 3, r = total & 7, xcd = orig & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int split = g.split_k > 1 ? g.split_k : 1;
  const int kchunk = ((g.K + split - 1) / split + BK - 1) / BK * BK;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(g.K, kbeg + kchunk);
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const int nk_full = kend > kbeg ? (kend - kbeg) / BK : 0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;
  const u16* __restrict__ A = reinterpret_cast<const u16*>(g.A);
  const u16* __restrict__ B = reinterpret_cast<const u16*>(g.B);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;

  uint4 ra[4], rb[4];
  auto fetch = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (kt < nk_full) {
      if (AKC) load_kc<false>(ra, A, g.lda, m0, k0, g.M, kend); else load_mc<false>(ra, A, g.lda, m0, k0, g.M, kend);
      if (BKC) load_kc<false>(rb, B, g.ldb, n0, k0, g.N, kend); else load_mc<false>(rb, B, g.ldb, n0, k0, g.N, kend);
    } else {
      if (AKC) load_kc<true>(ra, A, g.lda, m0, k0, g.M, kend); else load_mc<true>(ra, A, g.lda, m0, k0, g.M, kend);
      if (BKC) load_kc<true>(rb, B, g.ldb, n0, k0, g.N, kend); else load_mc<true>(rb, B, g.ldb, n0, k0, g.N, kend);
    }
  };
  auto stash = [&](int buf) {
    char* a_dst = lds + buf * STAGE;
    if (AKC) store_kc(a_dst, ra); else store_mc(a_dst, ra);
    if (BKC) store_kc(a_dst + TILE, rb); else store_mc(a_dst + TILE, rb);
  };
  if (nk > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* a_cur = lds + cur * STAGE;
    const char* b_cur = a_cur + TILE;
    const bool more = kt + 1 < nk;
    if (more) fetch(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = frag(a_cur, wm0 + a * 32 + li, ks, lh);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = frag(b_cur, wn0 + b * 32 + li, ks, lh);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    if (more) stash(cur ^ 1);
    __syncthreads();
  }

  // epilogue: lane owns column n0+wn0+b*32+li; register x is row (x&3)+8*(x>>2)+4*lh of the 32x32 tile
  const int epi = EPI == EPI_RUNTIME ? g.epilogue : EPI;
  float* C = g.C ? g.C + (size_t)blockIdx.z * (size_t)g.M * g.ldc : nullptr;
  u16* C16 = reinterpret_cast<u16*>(g.C16);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int col = n0 + wn0 + b * 32 + li;
    const bool col_ok = col < g.N;
    const float bias = (col_ok && (epi == ADDHIP_EPI_BIAS || epi == ADDHIP_EPI_BIAS_RELU)) ? g.bias[col] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int rbase = m0 + wm0 + a * 32 + 4 * lh;
      float mk[16];
      float cs = 0.f;
      const int cgroup = n0 + wn0 + b * 32;
      if (epi == ADDHIP_EPI_MASK) {
        if (g.mask_bits) {
#pragma unroll
          for (int x = 0; x < 16; ++x) {
            const int row = rbase + (x & 3) + 8 * (x >> 2);
            const unsigned wbits = (cgroup < g.N && row < g.M) ? g.mask_bits[(size_t)row * g.ldbits + (cgroup >> 5)] : 0u;
            mk[x] = ((wbits >> li) & 1u) ? 1.f : 0.f;
          }
        } else {
#pragma unroll
          for (int x = 0; x < 16; ++x) {
            const int row = rbase + (x & 3) + 8 * (x >> 2);
            mk[x] = (col_ok && row < g.M) ? g.mask[(size_t)row * g.ldmask + col] : 0.f;
          }
        }
      }
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int row = rbase + (x & 3) + 8 * (x >> 2);
        float v = g.alpha * acc[a][b][x] + bias;
        if (epi == ADDHIP_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
        if (epi == ADDHIP_EPI_MASK) v = mk[x] > 0.f ? v : 0.f;
        if (col_ok && row < g.M) {
          if (C) C[(size_t)row * g.ldc + col] = v;
          if (C16) C16[(size_t)row * g.ldc16 + col] = to_bf16(v);
          if (epi == ADDHIP_EPI_MASK) cs += v;
        }
        if (epi == ADDHIP_EPI_BIAS_RELU && g.relu_bits) {
          const unsigned long long pos = __ballot(col_ok && row < g.M && v > 0.f);
          if (li == 0 && row < g.M && cgroup < g.N) g.relu_bits[(size_t)row * g.ldbits + (cgroup >> 5)] = lh ? (unsigned)(pos >> 32) : (unsigned)pos;
        }
      }
      if (epi == ADDHIP_EPI_MASK && g.colsum) {
        cs += __shfl_xor(cs, 32, 64);
        if (lh == 0 && col_ok) atomicAdd(&g.colsum[col], cs);
      }
    }
  }
}

// fp32 -> bf16 (round to nearest even), row by row: dst[r*ld_dst + c] = bf16(src[r*ld_src + c]), cols % 4 == 0
__global__ void to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, long long rows, int cols, int ld_src, int ld_dst) {
  const int cq = cols >> 2;
  const long long n = rows * cq;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cq;
    const int c = (int)(i - r * cq) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
    const unsigned lo = (unsigned)to_bf16(v.x) | ((unsigned)to_bf16(v.y) << 16), hi = (unsigned)to_bf16(v.z) | ((unsigned)to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(dst + r * ld_dst + c) = make_uint2(lo, hi);
  }
}

// transposing variant: dst[c*ld_dst + r] = bf16(src[r*ld_src + c]) through a 32x33 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void to_bf16_t_kernel(const float* __restrict__ src, u16* __restrict__ dst, int rows, int cols, int ld_src, int ld_dst) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    tile[ty + 8 * j][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * ld_dst + r] = to_bf16(tile[tx][ty + 8 * j]);
  }
}

}  // namespace

namespace addhip {
// called by addhip_gemm_f32 (gemm.hip) after argument validation, for descriptors whose operands are stored as bf16
int gemm_bf16_dispatch(const addhip_gemm_t& g, hipStream_t st) {
  if (g.a_mean || g.a_std) return (set_error("gemm: fused normalisation is not built for bf16-stored operands"), -1);
  if (g.a_kcontig ? (g.K % 8 != 0 || g.lda % 8 != 0) : (g.M % 8 != 0 || g.lda % 8 != 0))
    return (set_error("gemm: bf16-stored A needs 16-byte chunks (K or M, and lda, multiples of 8)"), -1);
  if (g.b_kcontig ? (g.K % 8 != 0 || g.ldb % 8 != 0) : (g.N % 8 != 0 || g.ldb % 8 != 0))
    return (set_error("gemm: bf16-stored B needs 16-byte chunks (K or N, and ldb, multiples of 8)"), -1);
  if (!g.C && !g.C16) return (set_error("gemm: no output"), -1);
  if (g.split_k > 1 && (!g.C || g.C16)) return (set_error("gemm: split-K slabs are fp32"), -1);
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int split = g.split_k > 1 ? g.split_k : 1;
  dim3 grid(tiles_m * tiles_n, 1, split), block(256);
#define ADDHIP_LAUNCH(AK, BKc, EPI) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKc, EPI>), grid, block, 0, st, g, tiles_m, tiles_n)
  if (g.a_kcontig && g.b_kcontig) {
    if (g.epilogue == ADDHIP_EPI_BIAS_RELU) ADDHIP_LAUNCH(true, true, ADDHIP_EPI_BIAS_RELU);
    else if (g.epilogue == ADDHIP_EPI_MASK) ADDHIP_LAUNCH(true, true, ADDHIP_EPI_MASK);
    else ADDHIP_LAUNCH(true, true, EPI_RUNTIME);
  } else if (g.a_kcontig && !g.b_kcontig) {
    if (g.epilogue == ADDHIP_EPI_MASK) ADDHIP_LAUNCH(true, false, ADDHIP_EPI_MASK);
    else ADDHIP_LAUNCH(true, false, EPI_RUNTIME);
  } else if (!g.a_kcontig && g.b_kcontig) {
    ADDHIP_LAUNCH(false, true, EPI_RUNTIME);
  } else {
    if (g.epilogue == ADDHIP_EPI_NONE) ADDHIP_LAUNCH(false, false, ADDHIP_EPI_NONE);
    else ADDHIP_LAUNCH(false, false, EPI_RUNTIME);
  }
#undef ADDHIP_LAUNCH
  return check_launch("gemm_bf16_kernel");
}
}  // namespace addhip

extern "C" int addhip_to_bf16(const float* src, uint16_t* dst, int64_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, void* stream) {
  ADDHIP_REQUIRE(src && dst && rows > 0 && cols > 0 && cols % 4 == 0 && ld_src >= cols && ld_dst >= cols && ld_src % 4 == 0 && ld_dst % 4 == 0,
                 "to_bf16: bad arguments (cols and row strides must be multiples of 4)");
  ADDHIP_REQUIRE(aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0, "to_bf16: misaligned buffers");
  const long long n = (long long)rows * (cols / 4);
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)rows, cols, ld_src, ld_dst);
  return addhip::check_launch("to_bf16_kernel");
}

extern "C" int addhip_to_bf16_t(const float* src, uint16_t* dst, int32_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, void* stream) {
  ADDHIP_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "to_bf16_t: bad arguments");
  hipLaunchKernelGGL(to_bf16_t_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols, ld_src, ld_dst);
  return addhip::check_launch("to_bf16_t_kernel");
}

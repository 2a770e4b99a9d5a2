// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD)
// for the dense layers of the actor / critic / discriminator MLPs, forward and backward:
//   C[M,N] = alpha * sum_k A(m,k) * B(n,k)   with either operand k-contiguous or m/n-contiguous,
// fused epilogues (bias, bias+ReLU, ReLU-mask of the producing layer), optional fused input
// normalisation (x-mean)/std on A, and split-K partial slabs for the weight-gradient shapes.
//
// Tiling: 256 threads = 4 wavefronts; each wavefront owns a (BM/WM)x(BN/WN) sub-tile made of 32x32
// MFMA accumulators (64-lane layout: col = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)).
// K is walked in 16-deep LDS tiles, double buffered, with the next tile prefetched global->VGPR
// while the current one feeds the MFMAs.  k-contiguous operands sit in LDS as [row][16+4] and are
// read with one ds_read_b128 per 8 k (conflict-free: 20-dword stride spreads a 16-lane group over
// all 64 banks); m/n-contiguous operands sit as [k][rows+4] and are read with ds_read_b32.
// Workgroup ids are remapped so that the N-tiles sharing an A row-panel run on one XCD (one L2).
//
// Replaces: torch.nn.Linear forward/backward inside PPOModel.eval_actor/eval_critic
// (ppo_model.py:13-21), ADDModel.eval_disc (add_model.py:12-15) and their autograd.
#include "common.h"
#include "record.h"
#include "gemm_epilogue.h"
#include "gemm_dma.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// K depth of one LDS tile: 32 (two 36.9 KB stages -> exactly two 128x128 workgroups per CU, so the usual
// 512- and 1024-tile grids run as whole waves of workgroups over the 256 CUs) or 16 for the small tiles.
template <int ROWS, bool KC, int BK>
struct Tile {
  static constexpr int LDK = BK + 4;  // 20 / 36 dwords: a ds_read_b128 lane group covers all 64 banks
  static constexpr int SIZE = KC ? ROWS * LDK : BK * (ROWS + 4);
  static constexpr int F4 = ROWS * BK / 4;            // float4 per tile
  static constexpr int PER_THREAD = (F4 + 255) / 256;  // float4 per thread
  static constexpr int KQ = BK / 4;                    // float4 per k-contiguous row
};

// global -> registers for one operand tile (r0: first row of the tile in the M/N extent, k0: first k).
// GUARD=false is the interior fast path: the whole K range of the tile is in bounds, and rows beyond the
// matrix are clamped to the last valid row -- their products only reach C rows/columns that are never stored.
template <int ROWS, bool KC, int BK, bool GUARD, bool NORM>
__device__ __forceinline__ void load_tile(float4* reg, const float* __restrict__ P, int ld, int r0, int k0, int R, int kend,
                                          const float* __restrict__ mean, const float* __restrict__ stdv) {
  const int tid = threadIdx.x;
  using TT = Tile<ROWS, KC, BK>;
#pragma unroll
  for (int i = 0; i < TT::PER_THREAD; ++i) {
    const int f = tid + 256 * i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TT::F4 % 256 == 0 || f < TT::F4) {
      if (KC) {
        const int row = f / TT::KQ, kq = (f % TT::KQ) * 4;
        const int r = min(r0 + row, R - 1), k = k0 + kq;
        if (!GUARD || k < kend) {
          v = *reinterpret_cast<const float4*>(P + (size_t)r * ld + k);
          if (NORM) {  // Normalizer.normalize (normalizer.py:107-110)
            const float4 mu = *reinterpret_cast<const float4*>(mean + k);
            const float4 sd = *reinterpret_cast<const float4*>(stdv + k);
            v.x = (v.x - mu.x) / sd.x; v.y = (v.y - mu.y) / sd.y; v.z = (v.z - mu.z) / sd.z; v.w = (v.w - mu.w) / sd.w;
          }
        }
      } else {
        constexpr int RQ = ROWS / 4;
        const int kk = f / RQ, rq = (f - kk * RQ) * 4;
        const int r = min(r0 + rq, R - 4), k = k0 + kk;
        if (!GUARD || k < kend) v = *reinterpret_cast<const float4*>(P + (size_t)k * ld + r);
      }
    }
    reg[i] = v;
  }
}

template <int ROWS, bool KC, int BK>
__device__ __forceinline__ void store_tile(float* lds, const float4* reg) {
  const int tid = threadIdx.x;
  using TT = Tile<ROWS, KC, BK>;
#pragma unroll
  for (int i = 0; i < TT::PER_THREAD; ++i) {
    const int f = tid + 256 * i;
    if (TT::F4 % 256 == 0 || f < TT::F4) {
      if (KC) {
        const int row = f / TT::KQ, kq = (f % TT::KQ) * 4;
        *reinterpret_cast<float4*>(lds + row * TT::LDK + kq) = reg[i];
      } else {
        constexpr int RQ = ROWS / 4;
        const int kk = f / RQ, rq = (f - kk * RQ) * 4;
        *reinterpret_cast<float4*>(lds + kk * (ROWS + 4) + rq) = reg[i];
      }
    }
  }
}

// fragment of one 32-row block for the 8-deep k chunk starting at kk: element j feeds MFMA j
// (lane half h supplies k = kk + 4h + j; both operands use the same map so the products pair up)
template <int ROWS, bool KC, int BK>
__device__ __forceinline__ float4 read_frag(const float* lds, int row, int kk, int h) {
  if (KC) return *reinterpret_cast<const float4*>(lds + row * Tile<ROWS, KC, BK>::LDK + kk + 4 * h);
  const float* p = lds + (kk + 4 * h) * (ROWS + 4) + row;
  return make_float4(p[0], p[ROWS + 4], p[2 * (ROWS + 4)], p[3 * (ROWS + 4)]);
}

using addhip_epi::EPI_RUNTIME;

// EPI: compile-time epilogue (ADDHIP_EPI_*) or EPI_RUNTIME; NORM: fused (a-mean)/std on A
template <int BM, int BN, int WM, int WN, bool AKC, bool BKC, int BK, int EPI, bool NORM, bool SB>
__global__ __launch_bounds__(256, SB ? 4 : 2) void gemm_kernel(addhip_gemm_t g, int tiles_m, int tiles_n) {
  static_assert(WM * WN == 4, "4 wavefronts");
  static_assert(!NORM || AKC, "fused normalisation needs a k-contiguous A");
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 32, FN = TN / 32;
  static_assert(FM >= 1 && FN >= 1, "wave tile must hold a 32x32 accumulator");
  using TA = Tile<BM, AKC, BK>;
  using TB = Tile<BN, BKC, BK>;
  constexpr int STAGE = TA::SIZE + TB::SIZE;
  constexpr int EPI_FLOATS = 4 * addhip_epi::EpiBuf<FN>::WAVE_BYTES / 4;
  constexpr int NSTAGE = SB ? 1 : 2;
  __shared__ __attribute__((aligned(16))) float lds[NSTAGE * STAGE > EPI_FLOATS ? NSTAGE * STAGE : EPI_FLOATS];

  // XCD-aware remap (blocks b and b+8 share an XCD): give each XCD a contiguous run of tiles,
  // N-tile fastest, so the N-tiles of one A row-panel hit the same L2.  Bijective for any count.
  const int total = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q = total >> 3, r = total & 7, xcd = orig & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // split-K range of this z-slice
  const int split = g.split_k > 1 ? g.split_k : 1;
  const int kchunk = ((g.K + split - 1) / split + BK - 1) / BK * BK;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(g.K, kbeg + kchunk);
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const int nk_full = kend > kbeg ? (kend - kbeg) / BK : 0;  // tiles whose whole K range is in bounds

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave / WN) * TM, wn0 = (wave % WN) * TN;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; ++a)
#pragma unroll
    for (int b = 0; b < FN; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;

  float4 ra[TA::PER_THREAD], rb[TB::PER_THREAD];
  auto fetch = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (kt < nk_full) {
      load_tile<BM, AKC, BK, false, NORM>(ra, g.A, g.lda, m0, k0, g.M, kend, g.a_mean, g.a_std);
      load_tile<BN, BKC, BK, false, false>(rb, g.B, g.ldb, n0, k0, g.N, kend, nullptr, nullptr);
    } else {
      load_tile<BM, AKC, BK, true, NORM>(ra, g.A, g.lda, m0, k0, g.M, kend, g.a_mean, g.a_std);
      load_tile<BN, BKC, BK, true, false>(rb, g.B, g.ldb, n0, k0, g.N, kend, nullptr, nullptr);
    }
  };
  if (nk > 0) {
    fetch(0);
    store_tile<BM, AKC, BK>(lds, ra);
    store_tile<BN, BKC, BK>(lds + TA::SIZE, rb);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = SB ? 0 : (kt & 1);
    const float* a_cur = lds + cur * STAGE;
    const float* b_cur = a_cur + TA::SIZE;
    const bool more = kt + 1 < nk;
    if (more) fetch(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 8) {
      float4 fa[FM], fb[FN];
#pragma unroll
      for (int a = 0; a < FM; ++a) fa[a] = read_frag<BM, AKC, BK>(a_cur, wm0 + a * 32 + li, kk, lh);
#pragma unroll
      for (int b = 0; b < FN; ++b) fb[b] = read_frag<BN, BKC, BK>(b_cur, wn0 + b * 32 + li, kk, lh);
#pragma unroll
      for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
        }
    }
    if (SB) __syncthreads();  // one stage buffer: every wave is done reading it before the next tile goes in
    if (more) {
      store_tile<BM, AKC, BK>(lds + (SB ? 0 : (cur ^ 1)) * STAGE, ra);
      store_tile<BN, BKC, BK>(lds + (SB ? 0 : (cur ^ 1)) * STAGE + TA::SIZE, rb);
    }
    __syncthreads();
  }

  // epilogue (gemm_epilogue.h): every wave's block leaves through its private slice of the (now idle) stage buffers
  // (the loop's last barrier is behind every read of them)
  // (the 4-workgroups-per-CU form has no register to spare for the maximum: the dispatcher gives descriptors with amax_out the two-stage form)
  addhip_epi::gemm_epilogue<FM, FN, EPI, false, !SB>(g, acc, reinterpret_cast<char*>(lds) + wave * addhip_epi::EpiBuf<FN>::WAVE_BYTES, lane, m0 + wm0, n0 + wn0, blockIdx.z,
                                                      g.alpha);
}

template <int BM, int BN, int WM, int WN, int BK, bool SB = false>
int launch_cfg(const addhip_gemm_t& g, hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int split = g.split_k > 1 ? g.split_k : 1;
  dim3 grid(tiles_m * tiles_n, 1, split), block(256);
  const bool norm = g.a_mean != nullptr;
#define ADDHIP_LAUNCH(AK, BKc, EPI, NORM) \
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AK, BKc, BK, EPI, NORM, SB>), grid, block, 0, st, g, tiles_m, tiles_n)
  // hot combinations get a compile-time epilogue; everything else shares the run-time one
  if (g.a_kcontig && g.b_kcontig) {
    if (norm) {
      if (g.epilogue == ADDHIP_EPI_BIAS_RELU) ADDHIP_LAUNCH(true, true, ADDHIP_EPI_BIAS_RELU, true);
      else return (addhip::set_error("gemm: fused normalisation is only built for the bias+ReLU epilogue"), -1);
    } else if (g.epilogue == ADDHIP_EPI_BIAS_RELU) ADDHIP_LAUNCH(true, true, ADDHIP_EPI_BIAS_RELU, false);
    else if (g.epilogue == ADDHIP_EPI_MASK) ADDHIP_LAUNCH(true, true, ADDHIP_EPI_MASK, false);
    else ADDHIP_LAUNCH(true, true, EPI_RUNTIME, false);
  } else if (g.a_kcontig && !g.b_kcontig) {
    if (g.epilogue == ADDHIP_EPI_MASK) ADDHIP_LAUNCH(true, false, ADDHIP_EPI_MASK, false);
    else ADDHIP_LAUNCH(true, false, EPI_RUNTIME, false);
  } else if (!g.a_kcontig && g.b_kcontig) {
    ADDHIP_LAUNCH(false, true, EPI_RUNTIME, false);
  } else {
    if (g.epilogue == ADDHIP_EPI_NONE) ADDHIP_LAUNCH(false, false, ADDHIP_EPI_NONE, false);
    else ADDHIP_LAUNCH(false, false, EPI_RUNTIME, false);
  }
#undef ADDHIP_LAUNCH
  return addhip::check_launch("gemm_kernel");
}

// ------------------------------------------------------------------ a handful of rows (M <= 8, k-contiguous A)
// e.g. the discriminator's single zero-difference sample: a tile kernel would run 4 workgroups through the whole K loop.
// BKC: one wavefront per output column (lanes stride over k, wave reduction); !BKC: one thread per output column.
constexpr int SMALL_M = 8;
template <bool BKC>
__global__ __launch_bounds__(256) void gemm_small_m_kernel(addhip_gemm_t g) {
  float acc[SMALL_M];
#pragma unroll
  for (int m = 0; m < SMALL_M; ++m) acc[m] = 0.f;
  int n;
  bool writer;
  if (BKC) {
    n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= g.N) return;  // whole wave
    const float* b = g.B + (size_t)n * g.ldb;
    for (int k = lane * 4; k < g.K; k += 256) {  // K % 4 == 0 (checked by the caller)
      const float4 bv = *reinterpret_cast<const float4*>(b + k);
#pragma unroll
      for (int m = 0; m < SMALL_M; ++m)
        if (m < g.M) {
          const float4 av = *reinterpret_cast<const float4*>(g.A + (size_t)m * g.lda + k);
          acc[m] += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
        }
    }
#pragma unroll
    for (int m = 0; m < SMALL_M; ++m)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[m] += __shfl_xor(acc[m], o, 64);
    writer = lane == 0;
  } else {
    // 32 columns x 8 k-slices per workgroup; each thread walks its slice 8 rows at a time (8 independent loads in flight)
    __shared__ float part[8][SMALL_M][32];
    const int c = threadIdx.x & 31, ks = threadIdx.x >> 5;
    n = blockIdx.x * 32 + c;
    const int nn = min(n, g.N - 1);
    const int kper = (g.K + 7) / 8;
    const int k0 = ks * kper, k1 = min(g.K, k0 + kper);
    for (int k = k0; k < k1; k += 8) {
      float bv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = g.B[(size_t)min(k + j, g.K - 1) * g.ldb + nn];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (k + j < k1) {
#pragma unroll
          for (int m = 0; m < SMALL_M; ++m)
            if (m < g.M) acc[m] += g.A[(size_t)m * g.lda + k + j] * bv[j];
        }
    }
#pragma unroll
    for (int m = 0; m < SMALL_M; ++m) part[ks][m][c] = acc[m];
    __syncthreads();
    if (ks != 0 || n >= g.N) return;
#pragma unroll
    for (int m = 0; m < SMALL_M; ++m) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += part[q][m][c];
      acc[m] = t;
    }
    writer = true;
  }
  if (!writer) return;
  const int epi = g.epilogue;
  const float bias = (epi == ADDHIP_EPI_BIAS || epi == ADDHIP_EPI_BIAS_RELU) ? g.bias[n] : 0.f;
  float cs = 0.f, amx = 0.f;
#pragma unroll
  for (int m = 0; m < SMALL_M; ++m)
    if (m < g.M) {
      float v = g.alpha * acc[m] + bias;
      if (epi == ADDHIP_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
      if (epi == ADDHIP_EPI_MASK) v = g.mask[(size_t)m * g.ldmask + n] > 0.f ? v : 0.f;
      g.C[(size_t)m * g.ldc + n] = v;
      cs += v;
      amx = fmaxf(amx, fabsf(v));
    }
  if (epi == ADDHIP_EPI_MASK && g.colsum) atomicAdd(&g.colsum[n], cs);
  if (g.amax_out) atomicMax(&g.amax_out[blockIdx.x % ADDHIP_AMAX_SLOTS], __float_as_uint(amx));  // (one writer thread per column: few)
}

// ------------------------------------------------------------------ small reductions
__global__ void slab_reduce_kernel(const float* in, int slabs, long long stride, float* out, long long count, float scale, int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long step = (long long)gridDim.x * blockDim.x;
  for (; i < count; i += step) {
    float s = 0.f;
    for (int k = 0; k < slabs; ++k) s += in[k * stride + i];
    s *= scale;
    out[i] = accumulate ? out[i] + s : s;
  }
}
// the same, four elements per thread (count, stride % 4 == 0, 16-byte aligned buffers); sums in the same slab order
__global__ __launch_bounds__(256) void slab_reduce4_kernel(const float4* in, int slabs, long long stride4, float4* out, long long count4, float scale,
                                                           int accumulate) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < count4; i += step) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= slabs; k += 4) {
      const float4 a = in[k * stride4 + i], b = in[(k + 1) * stride4 + i], c = in[(k + 2) * stride4 + i], d = in[(k + 3) * stride4 + i];
      s.x = ((s.x + a.x) + b.x) + c.x + d.x; s.y = ((s.y + a.y) + b.y) + c.y + d.y;
      s.z = ((s.z + a.z) + b.z) + c.z + d.z; s.w = ((s.w + a.w) + b.w) + c.w + d.w;
    }
    for (; k < slabs; ++k) {
      const float4 a = in[k * stride4 + i];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
    if (accumulate) {
      const float4 o = out[i];
      s.x = o.x + s.x; s.y = o.y + s.y; s.z = o.z + s.z; s.w = o.w + s.w;
    }
    out[i] = s;
  }
}

// addhip_slab_reduce_pair: blockIdx.y == 0 the slab reduction (float4 form), == 1 the replica rows -> out2, read and cleared
__global__ __launch_bounds__(256) void slab_reduce_pair_kernel(const float4* in, int slabs, long long stride4, float4* out, long long count4, float scale, int accumulate,
                                                               float* in2, int rows2, int ld2, float* out2, int count2, int accumulate2, int clear2) {
  if (blockIdx.y == 0) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < count4; i += step) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int k = 0;
      for (; k + 4 <= slabs; k += 4) {  // (the summation order of slab_reduce4_kernel)
        const float4 a = in[k * stride4 + i], b = in[(k + 1) * stride4 + i], c = in[(k + 2) * stride4 + i], d = in[(k + 3) * stride4 + i];
        s.x = ((s.x + a.x) + b.x) + c.x + d.x; s.y = ((s.y + a.y) + b.y) + c.y + d.y;
        s.z = ((s.z + a.z) + b.z) + c.z + d.z; s.w = ((s.w + a.w) + b.w) + c.w + d.w;
      }
      for (; k < slabs; ++k) {
        const float4 a = in[k * stride4 + i];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      }
      s.x *= scale; s.y *= scale; s.z *= scale; s.w *= scale;
      if (accumulate) {
        const float4 o = out[i];
        s.x = o.x + s.x; s.y = o.y + s.y; s.z = o.z + s.z; s.w = o.w + s.w;
      }
      out[i] = s;
    }
    return;
  }
  // 64 columns per workgroup; wave w adds rows w, w + 4, w + 8, ... in increasing order (eight loads in flight), the four partial sums are
  // added in wave order: a FIXED order whatever the launch -- with agent.deterministic there are rows / 32 replica rows (512 at the headline's
  // minibatch), which one thread per column walking them all took 1.2 ms per launch for
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int n0 = blockIdx.x * 64; n0 < count2; n0 += gridDim.x * 64) {
    const int n = n0 + c;
    float s = 0.f;
    if (n < count2) {
      int r = w;
      for (; r + 28 < rows2; r += 32) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = in2[(size_t)(r + 4 * k) * ld2 + n];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s += v[k];
          if (clear2) in2[(size_t)(r + 4 * k) * ld2 + n] = 0.f;
        }
      }
      for (; r < rows2; r += 4) {
        s += in2[(size_t)r * ld2 + n];
        if (clear2) in2[(size_t)r * ld2 + n] = 0.f;
      }
    }
    part[w][c] = s;
    __syncthreads();
    if (w == 0 && n < count2) {
      const float t = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
      out2[n] = accumulate2 ? out2[n] + t : t;
    }
    __syncthreads();
  }
}

// column sums of a row-major [M,N] matrix: each block reduces a 64-column strip over a slice of rows
// into an LDS tile, then one atomicAdd per column per block (out must be pre-scaled/zeroed by the caller
// when accumulate==0: handled in the wrapper with a memset)
__global__ __launch_bounds__(256) void col_sum_kernel(const float* X, int M, int N, int ld, float* out, float scale, int rows_per_block) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N) {
    int r = r0 + w;
    for (; r + 28 < r1; r += 32) {  // eight loads in flight, added in row order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = X[(size_t)(r + 4 * k) * ld + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; r < r1; r += 4) s += X[(size_t)r * ld + c];
  }
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < N) {
    float t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
    atomicAdd(&out[c], t * scale);
  }
}

// fixed-order column sums (addhip_col_sum_ordered): slice blockIdx.y of the rows -> scratch[blockIdx.y][n] (the four waves of a workgroup
// are combined in wave order), then one thread per column adds the slices in slice order
__global__ __launch_bounds__(256) void col_sum_slices_kernel(const float* X, int M, int N, int ld, float* scratch, int rows_per_block) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (c < N) {
    int r = r0 + w;
    for (; r + 28 < r1; r += 32) {  // eight loads in flight, added in row order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = X[(size_t)(r + 4 * k) * ld + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; r < r1; r += 4) s += X[(size_t)r * ld + c];
  }
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < N) scratch[(size_t)blockIdx.y * N + c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void col_sum_combine_kernel(const float* scratch, int slices, int N, float* out, float scale, int accumulate) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < slices; ++k) s += scratch[(size_t)k * N + n];
  s *= scale;
  out[n] = accumulate ? out[n] + s : s;
}

}  // namespace

namespace addhip {
int gemm_split_dispatch(const addhip_gemm_t& g, int planes, hipStream_t st);       // gemm_split.hip
int gemm_bf16_dispatch(const addhip_dma::GemmGroup& grp, int count, hipStream_t st);  // gemm_bf16.hip
int gemm_x3_dispatch(const addhip_gemm_t& g, hipStream_t st);                          // gemm_x3.hip
}

namespace {

// argument checks shared by addhip_gemm_f32 and addhip_gemm_grouped; normalises alpha
int validate(addhip_gemm_t& g) {
  ADDHIP_REQUIRE(g.M > 0 && g.N > 0 && g.K > 0, "gemm: empty problem %d x %d x %d", g.M, g.N, g.K);
  ADDHIP_REQUIRE(g.A && g.B && (g.C || g.C16), "gemm: null operand");
  ADDHIP_REQUIRE(aligned16(g.A) && aligned16(g.B) && (g.lda % 4 == 0) && (g.ldb % 4 == 0), "gemm: operands must be 16-byte aligned with ld %% 4 == 0");
  if (g.a_kcontig) ADDHIP_REQUIRE(g.K % 4 == 0, "gemm: K must be a multiple of 4 for a k-contiguous A");
  else ADDHIP_REQUIRE(g.M % 4 == 0, "gemm: M must be a multiple of 4 for an m-contiguous A");
  if (g.b_kcontig) ADDHIP_REQUIRE(g.K % 4 == 0, "gemm: K must be a multiple of 4 for a k-contiguous B");
  else ADDHIP_REQUIRE(g.N % 4 == 0, "gemm: N must be a multiple of 4 for an n-contiguous B");
  ADDHIP_REQUIRE(g.epilogue >= ADDHIP_EPI_NONE && g.epilogue <= ADDHIP_EPI_MASK, "gemm: bad epilogue");
  if (g.epilogue == ADDHIP_EPI_BIAS || g.epilogue == ADDHIP_EPI_BIAS_RELU) ADDHIP_REQUIRE(g.bias, "gemm: bias missing");
  if (g.epilogue == ADDHIP_EPI_MASK) ADDHIP_REQUIRE(g.mask || g.mask_bits, "gemm: mask missing");
  if (g.mask_bits) ADDHIP_REQUIRE(g.epilogue == ADDHIP_EPI_MASK && (g.M > SMALL_M || g.operands_bf16) && g.ldbits * 32 >= g.N, "gemm: mask_bits need the MASK epilogue, M > 8 and ldbits >= ceil(N/32)");
  if (g.relu_bits) ADDHIP_REQUIRE(g.epilogue == ADDHIP_EPI_BIAS_RELU && (g.M > SMALL_M || g.operands_bf16) && g.ldbits * 32 >= g.N && g.split_k <= 1,
                                  "gemm: relu_bits need the BIAS_RELU epilogue, M > 8 and ldbits >= ceil(N/32)");
  if (g.split_k > 1) ADDHIP_REQUIRE(g.epilogue == ADDHIP_EPI_NONE, "gemm: split-K slabs take no epilogue");
  if (g.accumulate) ADDHIP_REQUIRE(g.epilogue == ADDHIP_EPI_NONE && !g.colsum, "gemm: accumulate takes no epilogue");
  if (g.a_mean || g.a_std) ADDHIP_REQUIRE(g.a_kcontig && g.a_mean && g.a_std, "gemm: fused normalisation needs a k-contiguous A and both mean/std");
  if (g.colsum) ADDHIP_REQUIRE(g.epilogue == ADDHIP_EPI_MASK && g.split_k <= 1, "gemm: colsum needs the MASK epilogue");
  if (g.colsum_replicas > 1) ADDHIP_REQUIRE(g.colsum && g.ldcs >= g.N && g.colsum_replicas <= 65536, "gemm: colsum_replicas (2..65536) need colsum and ldcs >= N");
  ADDHIP_REQUIRE(g.precision == ADDHIP_PREC_F32 || g.precision == ADDHIP_PREC_BF16 || g.precision == ADDHIP_PREC_BF16X2 ||
                     g.precision == ADDHIP_PREC_BF16X3 || g.precision == ADDHIP_PREC_F16X2, "gemm: bad precision");
  if (g.amax_out) ADDHIP_REQUIRE(g.split_k <= 1 && !g.accumulate && !g.operands_bf16, "gemm: amax_out tracks final results of fp32-operand GEMMs (not split-K slabs, not stored 16-bit operands)");
  ADDHIP_REQUIRE(g.C || g.split_k <= 1, "gemm: split-K slabs are fp32 (C)");
  if (g.operands_bf16) ADDHIP_REQUIRE(!g.accumulate, "gemm: accumulate is not built for bf16-stored operands");
  ADDHIP_REQUIRE(g.operands_bf16 == 0 || g.operands_bf16 == ADDHIP_STORE_BF16 || g.operands_bf16 == ADDHIP_STORE_BF16X3, "gemm: operands_bf16 is 0 or an ADDHIP_STORE_* format");
  ADDHIP_REQUIRE(g.c16_planes == 0 || g.c16_planes == ADDHIP_STORE_BF16 || g.c16_planes == ADDHIP_STORE_BF16X3, "gemm: c16_planes is 0 or an ADDHIP_STORE_* format");
  if (g.C16 && g.c16_planes == ADDHIP_STORE_BF16X3)
    ADDHIP_REQUIRE(g.operands_bf16 == ADDHIP_STORE_BF16X3 && g.N % 8 == 0 && g.ldc16 % 8 == 0 && aligned16(g.C16),
                   "gemm: a plane-storage C16 is written by GEMMs on plane-stored operands only (others: addhip_to_bf16x3 of the fp32 result); N and ldc16 multiples of 8, 16-byte aligned buffer");
  if (g.alpha == 0.0f) g.alpha = 1.0f;
  return 0;
}

// fp32 operands, 128x128 tiles, whole chip: the LDS-DMA kernel (gemm_dma.h).  Not for the fused input normalisation (the DMA
// cannot transform in flight: those launches -- first layers of the rollout / evaluation passes -- stay on gemm_kernel).
bool takes_dma_f32(const addhip_gemm_t& g, long long tiles128) {  // (amax_out: honoured by the register-staged and split kernels only)
  return g.precision == ADDHIP_PREC_F32 && !g.a_mean && tiles128 > 256 && !(g.hint & ADDHIP_GEMM_HINT_REG_STAGED) && !g.amax_out;
}
int launch_dma_f32(const addhip_dma::GemmGroup& grp, int count, hipStream_t st) {
  const addhip_gemm_t& g = grp.g[0];
  const int tiles_m = (g.M + 127) / 128, tiles_n = (g.N + 127) / 128, split = g.split_k > 1 ? g.split_k : 1;
  // one LDS stage x 4 workgroups per CU for the activation-streaming launches (forward, dX) that fill it -- the co-resident workgroups
  // cover each other's first fetch and write-out (16384x1024x1024: 272 us against 298 two-stage and 287 register-staged) -- two stages
  // x 2 per CU for the deep-K split-K weight gradients and for launches of fewer workgroups (profiles/r03_gemm_hint_sweep_fp32.log)
  bool one_stage = g.a_kcontig && split == 1 && (long long)tiles_m * tiles_n * count >= 768;
  if (g.hint & ADDHIP_GEMM_HINT_ONE_STAGE) one_stage = true;
  if (g.hint & ADDHIP_GEMM_HINT_TWO_STAGE) one_stage = false;
  if (one_stage) addhip_dma::launch_dma<addhip_dma::ElemF32, 1>(grp, count, tiles_m, tiles_n, split, st);
  else addhip_dma::launch_dma<addhip_dma::ElemF32, 0>(grp, count, tiles_m, tiles_n, split, st);
  return addhip::check_launch("gemm_dma_kernel<f32>");
}

// one fp32-operand problem (validated)
int dispatch_f32(const addhip_gemm_t& g, hipStream_t st) {
  if (g.M <= SMALL_M && g.a_kcontig && !g.a_mean && g.split_k <= 1 && !g.C16) {  // (the few-row kernel writes fp32 C only)
    if (g.b_kcontig) hipLaunchKernelGGL(gemm_small_m_kernel<true>, dim3((g.N + 3) / 4), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_small_m_kernel<false>, dim3((g.N + 31) / 32), dim3(256), 0, st, g);
    return addhip::check_launch("gemm_small_m_kernel");
  }
  // (32-deep K tiles for these latency-bound launches: 16-deep 26 / 38 us for the actor head's forward / weight gradient, 32-deep 18 / 27, 64-deep 19 / 31)
  if (g.N <= 32) return launch_cfg<128, 32, 4, 1, 32>(g, st);
  if (g.N <= 64) return launch_cfg<128, 64, 2, 2, 32>(g, st);
  // keep >= ~1 block per CU on the skinny rollout shapes
  const long long tiles128 = (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.split_k > 1 ? g.split_k : 1);
  // the bf16-MFMA paths pay off from half a chip of 128x128 tiles (measured: 16384x128x1024 77 -> 53 us, 4096x512x1024 64 -> 50 us)
  if (g.precision != ADDHIP_PREC_F32 && tiles128 >= 128 && tiles128 <= 256) return addhip::gemm_split_dispatch(g, g.precision, st);
  if (tiles128 <= 256) return launch_cfg<64, 128, 2, 2, 32>(g, st);  // (a bare chip of 128x128 tiles = one 4-wave workgroup per CU: two 64x128 ones overlap better)
  // N just past a multiple of 96 but far from one of 128 (the 272-wide first-layer weight gradient): 96-wide tiles waste
  // 6 % of their columns instead of 29 %
  const int waste128 = (g.N + 127) / 128 * 128 - g.N, waste96 = (g.N + 95) / 96 * 96 - g.N;
  const bool narrow = waste128 >= 64 && waste96 < 32 &&
                      (long long)((g.M + 127) / 128) * ((g.N + 95) / 96) * (g.split_k > 1 ? g.split_k : 1) >= 256;
  if (g.precision != ADDHIP_PREC_F32) return addhip::gemm_split_dispatch(g, g.precision, st);
  if (narrow) return launch_cfg<128, 96, 4, 1, 32>(g, st);
  if (takes_dma_f32(g, tiles128)) {
    addhip_dma::GemmGroup grp;
    grp.g[0] = g;
    return launch_dma_f32(grp, 1, st);
  }
  // register-staged 128x128 tiles (fused normalisation; ADDHIP_GEMM_HINT_REG_STAGED): launches with enough workgroups for 4 per CU
  // run one LDS stage (37 KB) x 4 workgroups per CU instead of two stages x 2
  const bool one_stage = g.amax_out ? false : (g.hint & ADDHIP_GEMM_HINT_ONE_STAGE) ? true : (g.hint & ADDHIP_GEMM_HINT_TWO_STAGE) ? false : tiles128 >= 512;
  if (one_stage) return launch_cfg<128, 128, 2, 2, 32, true>(g, st);
  return launch_cfg<128, 128, 2, 2, 32>(g, st);
}

}  // namespace

extern "C" int addhip_gemm_f32(const addhip_gemm_t* gp, void* stream) {
  ADDHIP_REQUIRE(gp, "null gemm descriptor");
  addhip_gemm_t g = *gp;
  if (int rc = validate(g)) return rc;
  if (addhip::recording())  // (record.h: the descriptor is kept by value, and kept visible to addhip_plan_call_gemms)
    return addhip::record_push("addhip_gemm_f32", [g](void* s) -> int { return addhip_gemm_f32(&g, s); }, &g, 1);
  hipStream_t st = (hipStream_t)stream;
  if (g.operands_bf16 == ADDHIP_STORE_BF16X3) return addhip::gemm_x3_dispatch(g, st);
  if (g.operands_bf16) {
    addhip_dma::GemmGroup grp;
    grp.g[0] = g;
    return addhip::gemm_bf16_dispatch(grp, 1, st);
  }
  return dispatch_f32(g, st);
}

extern "C" int addhip_gemm_grouped(const addhip_gemm_t* problems, int32_t count, void* stream) {
  ADDHIP_REQUIRE(problems && count >= 1 && count <= ADDHIP_GEMM_MAX_GROUP, "gemm_grouped: 1..%d problems", ADDHIP_GEMM_MAX_GROUP);
  addhip_dma::GemmGroup grp;
  for (int i = 0; i < count; ++i) {
    grp.g[i] = problems[i];
    if (int rc = validate(grp.g[i])) return rc;
    const addhip_gemm_t &a = grp.g[0], &b = grp.g[i];
    ADDHIP_REQUIRE(a.M == b.M && a.N == b.N && a.K == b.K && a.a_kcontig == b.a_kcontig && a.b_kcontig == b.b_kcontig && a.epilogue == b.epilogue &&
                       a.split_k == b.split_k && a.precision == b.precision && a.operands_bf16 == b.operands_bf16 && a.c16_planes == b.c16_planes && a.accumulate == b.accumulate &&
                       (a.a_mean != nullptr) == (b.a_mean != nullptr) && a.hint == b.hint,
                   "gemm_grouped: problem %d differs from problem 0 in shape, layout, epilogue, split, precision or storage", i);
  }
  if (addhip::recording())
    return addhip::record_push("addhip_gemm_grouped", [grp, count](void* s) -> int { return addhip_gemm_grouped(grp.g, count, s); }, grp.g, count);
  hipStream_t st = (hipStream_t)stream;
  const addhip_gemm_t& g = grp.g[0];
  if (count > 1) {
    // one launch over all problems where the shape takes the 128x128 LDS-DMA kernel ...
    const long long tiles128 = (long long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.split_k > 1 ? g.split_k : 1);
    if (g.operands_bf16 == ADDHIP_STORE_BF16 && !(g.hint & ADDHIP_GEMM_HINT_BIG_TILE)) return addhip::gemm_bf16_dispatch(grp, count, st);
    const int waste128 = (g.N + 127) / 128 * 128 - g.N, waste96 = (g.N + 95) / 96 * 96 - g.N;
    const bool narrow = waste128 >= 64 && waste96 < 32;
    if (!g.operands_bf16 && g.M > SMALL_M && g.N > 64 && !narrow && takes_dma_f32(g, tiles128 * count)) return launch_dma_f32(grp, count, st);
    // ... and one by one otherwise (same results; the group is then only a convenience)
    for (int i = 0; i < count; ++i) {
      const int rc = addhip_gemm_f32(&problems[i], stream);
      if (rc) return rc;
    }
    return 0;
  }
  return addhip_gemm_f32(problems, stream);
}

extern "C" int addhip_slab_reduce(const float* in, int32_t slabs, int64_t slab_stride, float* out, int64_t count, float scale,
                                  int32_t accumulate, void* stream) {
  ADDHIP_REQUIRE(in && out && slabs > 0 && count > 0, "slab_reduce: bad arguments");
  ADDHIP_RECORDABLE(addhip_slab_reduce, in, slabs, slab_stride, out, count, scale, accumulate);
  if (count % 4 == 0 && slab_stride % 4 == 0 && aligned16(in) && aligned16(out)) {
    long long b4 = (count / 4 + 255) / 256;
    if (b4 > 2048) b4 = 2048;
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(in), slabs,
                       (long long)slab_stride / 4, reinterpret_cast<float4*>(out), (long long)count / 4, scale, accumulate);
    return addhip::check_launch("slab_reduce4_kernel");
  }
  int blocks = (int)((count + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, slabs, (long long)slab_stride, out,
                     (long long)count, scale, accumulate);
  return addhip::check_launch("slab_reduce_kernel");
}

extern "C" int addhip_slab_reduce_pair(const float* in, int32_t slabs, int64_t slab_stride, float* out, int64_t count, float scale, int32_t accumulate,
                                       float* in2, int32_t rows2, int32_t ld2, float* out2, int32_t count2, int32_t accumulate2, int32_t clear2, void* stream) {
  ADDHIP_REQUIRE(in && out && slabs > 0 && count > 0, "slab_reduce_pair: bad arguments");
  ADDHIP_REQUIRE(count % 4 == 0 && slab_stride % 4 == 0 && aligned16(in) && aligned16(out), "slab_reduce_pair: the slab reduction takes 16-byte aligned buffers, count and stride %% 4 == 0");
  ADDHIP_REQUIRE(in2 && out2 && rows2 > 0 && rows2 <= 65536 && count2 > 0 && ld2 >= count2, "slab_reduce_pair: bad second reduction");
  ADDHIP_RECORDABLE(addhip_slab_reduce_pair, in, slabs, slab_stride, out, count, scale, accumulate, in2, rows2, ld2, out2, count2, accumulate2, clear2);
  long long b4 = (count / 4 + 255) / 256;
  if (b4 > 2048) b4 = 2048;
  const long long b2 = (count2 + 63) / 64;
  if (b4 < b2) b4 = b2;
  hipLaunchKernelGGL(slab_reduce_pair_kernel, dim3((unsigned)b4, 2), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(in), slabs,
                     (long long)slab_stride / 4, reinterpret_cast<float4*>(out), (long long)count / 4, scale, accumulate, in2, rows2, ld2, out2, count2, accumulate2,
                     clear2);
  return addhip::check_launch("slab_reduce_pair_kernel");
}

extern "C" int addhip_fill_zero(float* p, int64_t count, void* stream) {
  ADDHIP_REQUIRE(p && count > 0, "fill_zero: bad arguments");
  ADDHIP_RECORDABLE(addhip_fill_zero, p, count);
  ADDHIP_HIP(hipMemsetAsync(p, 0, sizeof(float) * (size_t)count, (hipStream_t)stream));
  return 0;
}

extern "C" int addhip_col_sum_ordered(const float* X, int32_t M, int32_t N, int32_t ld, float* out, float scale, int32_t accumulate, float* scratch, void* stream) {
  ADDHIP_REQUIRE(X && out && scratch && M > 0 && N > 0 && ld >= N, "col_sum_ordered: bad arguments");
  ADDHIP_RECORDABLE(addhip_col_sum_ordered, X, M, N, ld, out, scale, accumulate, scratch);
  hipStream_t st = (hipStream_t)stream;
  const int rows_per_block = (M + ADDHIP_ORDERED_BLOCKS - 1) / ADDHIP_ORDERED_BLOCKS;
  hipLaunchKernelGGL(col_sum_slices_kernel, dim3((N + 63) / 64, ADDHIP_ORDERED_BLOCKS), dim3(256), 0, st, X, M, N, ld, scratch, rows_per_block);
  hipLaunchKernelGGL(col_sum_combine_kernel, dim3((N + 255) / 256), dim3(256), 0, st, scratch, ADDHIP_ORDERED_BLOCKS, N, out, scale, accumulate);
  return addhip::check_launch("col_sum_ordered");
}

extern "C" int addhip_col_sum(const float* X, int32_t M, int32_t N, int32_t ld, float* out, float scale, int32_t accumulate, void* stream) {
  ADDHIP_REQUIRE(X && out && M > 0 && N > 0 && ld >= N, "col_sum: bad arguments");
  ADDHIP_RECORDABLE(addhip_col_sum, X, M, N, ld, out, scale, accumulate);
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) ADDHIP_HIP(hipMemsetAsync(out, 0, sizeof(float) * N, st));
  int strips = (N + 63) / 64;
  int ysplit = 1;
  while (strips * ysplit < 512 && M / (ysplit * 2) >= 64) ysplit *= 2;
  int rows_per_block = (M + ysplit - 1) / ysplit;
  hipLaunchKernelGGL(col_sum_kernel, dim3(strips, ysplit), dim3(256), 0, st, X, M, N, ld, out, scale, rows_per_block);
  return addhip::check_launch("col_sum_kernel");
}

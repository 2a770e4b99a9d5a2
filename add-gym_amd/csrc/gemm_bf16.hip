// GEMM on bf16 operands STORED as bf16 in HBM (agent.matmul_precision = bf16: bf16 hidden activations, pre-activation gradients
// and weight shadows; fp32 master weights, fp32 accumulation, fp32 or bf16 results):
//   C[M,N] = alpha * sum_k A(m,k) * B(n,k),  A, B bf16, either operand k-contiguous or m/n-contiguous,
// same epilogues, ReLU sign bits, fused bias-gradient column sums and split-K slabs as gemm.hip / gemm_split.hip.
//
// Unlike gemm_split.hip (fp32 operands split into bf16 planes on their way into LDS, 16-deep stages) nothing is converted
// here.  The 128x128-tile kernel is gemm_dma.h's gemm_dma_kernel<ElemBF16, ...> (shared with the fp32 path: tiles by LDS-DMA,
// bank-swizzled through the source addresses; m/n-contiguous operands are NOT transposed on the way in: their fragments are read
// with ds_read_b64_tr_b16); this file adds the 256x256 ring kernel for deep-K shapes, the dispatcher and the bf16 conversions.
//
// Replaces (in bf16 mode): the torch.nn.Linear forward/backward of PPOModel / ADDModel (ppo_model.py:13-21, add_model.py:12-15).
#include "common.h"
#include "record.h"
#include "gemm_epilogue.h"
#include "gemm_dma.h"
#include "planes.h"

namespace {

typedef unsigned short u16;
using namespace addhip_dma;  // the 128x128 LDS-DMA kernel (gemm_dma_kernel<ElemBF16, ...>), its stagers and fragment readers

// fp32 -> bf16, round to nearest even (no NaN special-casing: the callers' values are finite)
__device__ __forceinline__ u16 to_bf16(float v) {
  const unsigned u = __float_as_uint(v);
  return (u16)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4) of 128x64, one workgroup per CU.  Why it exists: every K stage of these GEMMs holds rows that
// no CU has touched yet (the minibatch operand is streamed once), so a stage's DMA takes an HBM round trip, ~1.5 us under
// load (measured: a launch that only loads runs as long as one that only computes), and the L2 -> LDS path serves a CU about
// 0.3 requests per clock whether they are 128-byte lines or halves.  The 128x128 kernel can only put other workgroups' MFMAs
// under that latency and needs 64 B/clk/CU; this one needs 32 B/clk/CU, asks for whole lines only, and keeps its own DMA
// several phases ahead of the MFMAs behind counted vmcnt waits and raw barriers (no vmcnt(0) inside the loop).
//
// A 64-deep K tile arrives as four 16 KB units of 128 rows x 64 k (whole 128-byte lines of a k-contiguous operand):
//   u0 = AQ0, u1 = BQ0, u2 = BQ1, u3 = AQ1;  AQi = the i-th 64-row half of both wave-rows' 128 rows, BQj = the j-th 32-column
//   half of all four wave-columns' 64 columns,
// and is consumed in four phases, one 64x32 quadrant of every wave's block each: (A half, B half) = (0,0) (0,1) (1,0) (1,1),
// 8 MFMAs per wave and phase.  Phase f (counting through all tiles) issues unit f+AHEAD into ring slot (f+AHEAD) % NS, waits
// until units <= f+2 have landed (this wave's share; the barrier covers the others'), passes the barrier, requests the
// fragments the NEXT phase needs from LDS, then issues this phase's MFMAs on fragments requested one phase ago.
// Slot reuse: unit v is read in phase v-1 or v-2; those reads have returned before their wave reaches the next barrier, so
// a slot may be overwritten two phases later: AHEAD = NS - 1.
constexpr int Q_NS = 8, Q_AHEAD = Q_NS - 1, Q_UNIT = 16384;

// DMA addressing of one operand's units: 16 pieces per unit, 2 per wave
template <bool KC, int GS>  // GS = rows of one wave's share in a unit (A: 64, B: 32); the shares of consecutive waves lie 2*GS apart in the tile
struct QStager {
  unsigned off[2];
  __device__ __forceinline__ void init(int wave, int lane, int ld) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = 2 * wave + i;
      if (KC) {
        const int lr = 8 * piece + (lane >> 3), c = (lane & 7) ^ kc_swz(lr);
        const int tr = (lr / GS) * (2 * GS) + (lr % GS);
        off[i] = 2u * ((unsigned)tr * (unsigned)ld + 8u * c);
      } else {
        const int kr = 4 * piece + (lane >> 4), ch = (lane & 15) ^ ElemBF16::mc_swz(kr);
        const int lm = 8 * ch, tm = (lm / GS) * (2 * GS) + (lm % GS);
        off[i] = 2u * ((unsigned)kr * (unsigned)ld + (unsigned)tm);
      }
    }
  }
  __device__ __forceinline__ void issue(const char* src, char* slot, int wave) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(src + off[i]), (lptr_t)(slot + (2 * wave + i) * 1024), 16, 0, 0);
  }
};

template <bool AKC, bool BKC, int EPI>
__global__ __launch_bounds__(512, 1) void gemm_bf16_q_kernel(addhip_gemm_t g, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) char lds[Q_NS * Q_UNIT];
  static_assert(Q_NS * Q_UNIT >= 8 * EpiBuf<2>::WAVE_BYTES, "the epilogue buffers must fit");
  const int total = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q = total >> 3, r = total & 7, xcd = orig & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;

  const int split = g.split_k > 1 ? g.split_k : 1;
  const int kchunk = ((g.K + split - 1) / split + 63) / 64 * 64;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(g.K, kbeg + kchunk);
  const int nk = kend > kbeg ? (kend - kbeg) / 64 : 0;  // the host sends only K % 64 == 0 here
  const int U = 4 * nk;

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 31, lh = lane >> 5;

  const char* srcA = reinterpret_cast<const char*>(g.A) + 2 * (AKC ? (size_t)m0 * g.lda + kbeg : (size_t)kbeg * g.lda + m0);
  const char* srcB = reinterpret_cast<const char*>(g.B) + 2 * (BKC ? (size_t)n0 * g.ldb + kbeg : (size_t)kbeg * g.ldb + n0);
  const size_t tileA = 2 * (AKC ? (size_t)64 : (size_t)64 * g.lda), tileB = 2 * (BKC ? (size_t)64 : (size_t)64 * g.ldb);   // next K tile
  const size_t halfA = 2 * (AKC ? (size_t)64 * g.lda : (size_t)64), halfB = 2 * (BKC ? (size_t)32 * g.ldb : (size_t)32);   // second half
  QStager<AKC, 64> sa;
  QStager<BKC, 32> sb;
  sa.init(wave, lane, g.lda);
  sb.init(wave, lane, g.ldb);
  typename FragSel<ElemBF16, AKC, 128, 2>::type fa_;
  typename FragSel<ElemBF16, BKC, 128, 1>::type fb_;
  if constexpr (AKC) fa_.init(wm * 64, li, lh); else fa_.init(wm * 64, lane);
  if constexpr (BKC) fb_.init(wn * 32, li, lh); else fb_.init(wn * 32, lane);

  f32x16 acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;

  auto issue_unit = [&](int u) {  // u < U
    const int t = u >> 2, w = u & 3;
    char* slot = lds + (u % Q_NS) * Q_UNIT;
    if (w == 0) sa.issue(srcA + t * tileA, slot, wave);
    else if (w == 3) sa.issue(srcA + t * tileA + halfA, slot, wave);
    else sb.issue(srcB + t * tileB + (w == 2 ? halfB : 0), slot, wave);
  };
  // units <= f+2 landed, as far as this wave's own loads go: at most min(AHEAD, U-1-f) - 2 later units (2 loads each) outstanding
  auto wait_landed = [&](int f) {
    const int later = min(Q_AHEAD, U - 1 - f) - 2;
    if (later >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (later == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(Q_AHEAD - 2 == 5, "wait_landed's immediates are written for AHEAD = 7");

  bf16x8 fa[2][4][2];  // [A half][k step][block]
  bf16x8 fb[2][4];     // [B half][k step]
  auto read_a = [&](int i, int u) {
    const char* slot = lds + (u % Q_NS) * Q_UNIT;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[i][ks][a] = fa_.get(slot, a, ks);
  };
  auto read_b = [&](int j, int u) {
    const char* slot = lds + (u % Q_NS) * Q_UNIT;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[j][ks] = fb_.get(slot, 0, ks);
  };

  if (nk > 0) {
#pragma unroll
    for (int u = 0; u < Q_AHEAD; ++u)
      if (u < U) issue_unit(u);
    wait_landed(-1);  // units 0 and 1
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_a(0, 0);
    read_b(0, 1);
  }
  for (int t = 0; t < nk; ++t) {
    const int f0 = 4 * t;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int f = f0 + p;
      if (f + Q_AHEAD < U) issue_unit(f + Q_AHEAD);
      wait_landed(f);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // fragments of the next phase (their units landed before the barrier just passed)
      if (p == 0) read_b(1, f0 + 2);
      if (p == 1) read_a(1, f0 + 3);
      if (p == 3 && t + 1 < nk) {
        read_a(0, f0 + 4);
        read_b(0, f0 + 5);
      }
      const int i = p >> 1, j = p & 1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[2 * i + a][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][ks][a], fb[j][ks], acc[2 * i + a][j], 0, 0, 0);
    }
  }
  __syncthreads();
  gemm_epilogue<4, 2, EPI>(g, acc, lds + wave * EpiBuf<2>::WAVE_BYTES, lane, m0 + wm * 128, n0 + wn * 64, blockIdx.z);
}

// fp32 -> bf16 (round to nearest even), row by row: dst[r*ld_dst + c] = bf16(src[r*ld_src + c]), cols % 4 == 0
template <bool X3>
__global__ void to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, long long rows, int cols, int ld_src, int ld_dst) {
  const int cq = cols >> 2;
  const long long n = rows * cq;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cq;
    const int c = (int)(i - r * cq) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
    if (X3) {  // plane storage (planes.h): the exact 3-way split
      addhip_planes::store4(dst + 3 * r * ld_dst, c, v);
      continue;
    }
    const unsigned lo = (unsigned)to_bf16(v.x) | ((unsigned)to_bf16(v.y) << 16), hi = (unsigned)to_bf16(v.z) | ((unsigned)to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(dst + r * ld_dst + c) = make_uint2(lo, hi);
  }
}

// the same behind Normalizer.normalize: dst = bf16((src - mean[c]) / std[c])   (a true division, as normalizer.py:107-110 and the GEMMs' fused form)
__global__ void normalize_to_bf16_kernel(const float* __restrict__ src, const float* __restrict__ mean, const float* __restrict__ stdv, u16* __restrict__ dst,
                                         long long rows, int cols, int ld_src, int ld_dst) {
  const int cq = cols >> 2;
  const long long n = rows * cq;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cq;
    const int c = (int)(i - r * cq) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + r * ld_src + c);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c), sd = *reinterpret_cast<const float4*>(stdv + c);
    const unsigned lo = (unsigned)to_bf16((v.x - mu.x) / sd.x) | ((unsigned)to_bf16((v.y - mu.y) / sd.y) << 16);
    const unsigned hi = (unsigned)to_bf16((v.z - mu.z) / sd.z) | ((unsigned)to_bf16((v.w - mu.w) / sd.w) << 16);
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

// flat convert + the transposed copies of up to ADDHIP_SHADOW_MAX_MATS matrices, one launch: blocks [0, flat_blocks) stride over the
// flat buffer, the rest are 32x32 transposition tiles, matrix by matrix (tile_end = running block count)
struct ShadowMats {
  long long offset[ADDHIP_SHADOW_MAX_MATS];
  int rows[ADDHIP_SHADOW_MAX_MATS], cols[ADDHIP_SHADOW_MAX_MATS], tile_end[ADDHIP_SHADOW_MAX_MATS];
  int n;
};
template <bool X3>
__global__ __launch_bounds__(256) void shadow_refresh_kernel(const float* __restrict__ params, u16* __restrict__ flat16, u16* __restrict__ trans16,
                                                             long long count, int flat_blocks, ShadowMats mats) {
  __shared__ float tile[32][33];
  if ((int)blockIdx.x < flat_blocks) {
    const long long n4 = count >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)flat_blocks * 256) {
      const float4 v = reinterpret_cast<const float4*>(params)[i];
      if (X3) {  // (count % 8 == 0)
        addhip_planes::store4_flat(flat16, 4 * i, v);
        continue;
      }
      reinterpret_cast<uint2*>(flat16)[i] =
          make_uint2((unsigned)to_bf16(v.x) | ((unsigned)to_bf16(v.y) << 16), (unsigned)to_bf16(v.z) | ((unsigned)to_bf16(v.w) << 16));
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) flat16[(n4 << 2) + threadIdx.x] = to_bf16(params[(n4 << 2) + threadIdx.x]);
    return;
  }
  int t = blockIdx.x - flat_blocks, mi = 0;
  while (mi + 1 < mats.n && t >= mats.tile_end[mi]) ++mi;
  if (mi > 0) t -= mats.tile_end[mi - 1];
  const int rows = mats.rows[mi], cols = mats.cols[mi];
  const int tcols = (cols + 31) / 32;
  const int r0 = (t / tcols) * 32, c0 = (t % tcols) * 32;
  const float* src = params + mats.offset[mi];
  u16* dst = trans16 + (X3 ? 3 : 1) * mats.offset[mi];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    tile[ty + 8 * j][tx] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  if (X3) {  // transposed row c = rows values (rows % 8 == 0): a thread writes 4 of them into the three planes
    const int c = c0 + (threadIdx.x >> 3), rq = (threadIdx.x & 7) * 4, r = r0 + rq;
    if (c < cols && r < rows)
      addhip_planes::store4(dst + 3 * (size_t)c * rows, r, make_float4(tile[rq][c - c0], tile[rq + 1][c - c0], tile[rq + 2][c - c0], tile[rq + 3][c - c0]));
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;
    if (r < rows && c < cols) dst[(size_t)c * rows + r] = to_bf16(tile[tx][ty + 8 * j]);
  }
}

}  // namespace

namespace addhip {
// called by the GEMM entry points (gemm.hip) after argument validation, for descriptors whose operands are stored as bf16;
// count > 1: equal-shaped problems of one grouped launch
int gemm_bf16_dispatch(const addhip_dma::GemmGroup& grp, int count, hipStream_t st) {
  const addhip_gemm_t& g = grp.g[0];
  for (int i = 0; i < count; ++i) {
    const addhip_gemm_t& p = grp.g[i];
    if (p.a_mean || p.a_std) return (set_error("gemm: fused normalisation is not built for bf16-stored operands"), -1);
    if (p.a_kcontig ? (p.K % 8 != 0 || p.lda % 8 != 0) : (p.M % 8 != 0 || p.lda % 8 != 0))
      return (set_error("gemm: bf16-stored A needs 16-byte chunks (K or M, and lda, multiples of 8)"), -1);
    if (p.b_kcontig ? (p.K % 8 != 0 || p.ldb % 8 != 0) : (p.N % 8 != 0 || p.ldb % 8 != 0))
      return (set_error("gemm: bf16-stored B needs 16-byte chunks (K or N, and ldb, multiples of 8)"), -1);
    if (!p.C && !p.C16) return (set_error("gemm: no output"), -1);
    if (p.split_k > 1 && (!p.C || p.C16)) return (set_error("gemm: split-K slabs are fp32"), -1);
  }
  const int split = g.split_k > 1 ? g.split_k : 1;
  // 128x128 tiles: a single LDS stage x 4 (3 with a transposed operand) workgroups per CU when the launch has the workgroups to
  // half-fill that, double-buffered stages x 2 workgroups per CU if not (profiles/r03_gemm_hint_sweep_bf16.log: 1024-tile launches
  // 8-20 % faster single-buffered, the 512-528-workgroup weight gradients 8-20 %, 512-workgroup forward launches the same).  256x256 tiles (gemm_bf16_q_kernel, one workgroup per CU) only for
  // whole-tile shapes that fill the chip AND are deep in K: at 4096^3 it runs 1040 TFLOP/s on random operands, but with the
  // 1024-deep K of the training step's launches its prologue and 256x256 write-out are not amortised (725 TFLOP/s isolated, the
  // same as the 128x128 kernel) and, holding a CU's LDS alone, it keeps the other streams' launches off the CU (update phase
  // 47.8 vs 46.3 ms) -- so the step never selects it.  addhip_gemm_t.hint forces it on (eligible shapes) / off.
  const long long t256 = (long long)(g.M / 256) * (g.N / 256);
  const bool eligible = count == 1 && g.M % 256 == 0 && g.N % 256 == 0 && g.K % 64 == 0;
  bool big = eligible && t256 * split >= 192 && g.K / split >= 2048;
  // (round 4, re-measured: alone on the chip and on random operands the forward launches -- both operands k-contiguous, bias + ReLU -- run
  //  8-13 % faster on the 256x256 tiles (16384x1024x1024: 40.8 vs 45.7 us; 65536 rows: 161 vs 182; profiles/r04_bf16_tile_sweep.log), the dX
  //  and weight-gradient launches do not; given to the step it was again SLOWER end to end -- 2.58 M vs 2.67 M env-steps/s at 4096 envs,
  //  3.15 M vs 3.21 M at 16 384 -- for the reason above, so the rule stays as it was)
  if (g.hint & ADDHIP_GEMM_HINT_BIG_TILE) big = eligible;
  if (g.hint & ADDHIP_GEMM_HINT_NO_BIG_TILE) big = false;
  const int BM = big ? 256 : 128, BN = BM;
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  if (!big) {
    bool one_stage = (long long)tiles_m * tiles_n * split * count >= 512;
    if (g.hint & ADDHIP_GEMM_HINT_ONE_STAGE) one_stage = true;
    if (g.hint & ADDHIP_GEMM_HINT_TWO_STAGE) one_stage = false;
    if (one_stage) addhip_dma::launch_dma<addhip_dma::ElemBF16, 1>(grp, count, tiles_m, tiles_n, split, st);
    else addhip_dma::launch_dma<addhip_dma::ElemBF16, 0>(grp, count, tiles_m, tiles_n, split, st);
    return check_launch("gemm_dma_kernel<bf16>");
  }
  dim3 grid(tiles_m * tiles_n, 1, split), block(512);
#define ADDHIP_LAUNCH(AK, BKc, EPI) hipLaunchKernelGGL((gemm_bf16_q_kernel<AK, BKc, EPI>), grid, block, 0, st, g, tiles_m, tiles_n)
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
  return check_launch("gemm_bf16_q_kernel");
}
}  // namespace addhip

extern "C" int addhip_to_bf16(const float* src, uint16_t* dst, int64_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, void* stream) {
  ADDHIP_REQUIRE(src && dst && rows > 0 && cols > 0 && cols % 4 == 0 && ld_src >= cols && ld_dst >= cols && ld_src % 4 == 0 && ld_dst % 4 == 0,
                 "to_bf16: bad arguments (cols and row strides must be multiples of 4)");
  ADDHIP_REQUIRE(aligned16(src) && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0, "to_bf16: misaligned buffers");
  ADDHIP_RECORDABLE(addhip_to_bf16, src, dst, rows, cols, ld_src, ld_dst);
  const long long n = (long long)rows * (cols / 4);
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(to_bf16_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)rows, cols, ld_src, ld_dst);
  return addhip::check_launch("to_bf16_kernel");
}

extern "C" int addhip_normalize_to_bf16(const float* src, const float* mean, const float* stdv, uint16_t* dst, int64_t rows, int32_t cols, int32_t ld_src,
                                        int32_t ld_dst, void* stream) {
  ADDHIP_REQUIRE(src && mean && stdv && dst && rows > 0 && cols > 0 && cols % 4 == 0 && ld_src >= cols && ld_dst >= cols && ld_src % 4 == 0 && ld_dst % 4 == 0,
                 "normalize_to_bf16: bad arguments (cols and row strides must be multiples of 4)");
  ADDHIP_REQUIRE(aligned16(src) && aligned16(mean) && aligned16(stdv) && (reinterpret_cast<uintptr_t>(dst) & 7u) == 0, "normalize_to_bf16: misaligned buffers");
  ADDHIP_RECORDABLE(addhip_normalize_to_bf16, src, mean, stdv, dst, rows, cols, ld_src, ld_dst);
  const long long n = (long long)rows * (cols / 4);
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(normalize_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, mean, stdv, dst, (long long)rows, cols, ld_src, ld_dst);
  return addhip::check_launch("normalize_to_bf16_kernel");
}

extern "C" int addhip_to_bf16x3(const float* src, uint16_t* dst, int64_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, void* stream) {
  ADDHIP_REQUIRE(src && dst && rows > 0 && cols > 0 && cols % 8 == 0 && ld_src >= cols && ld_dst >= cols && ld_src % 4 == 0 && ld_dst % 8 == 0,
                 "to_bf16x3: bad arguments (cols and ld_dst multiples of 8, ld_src of 4)");
  ADDHIP_REQUIRE(aligned16(src) && aligned16(dst), "to_bf16x3: misaligned buffers");
  ADDHIP_RECORDABLE(addhip_to_bf16x3, src, dst, rows, cols, ld_src, ld_dst);
  const long long n = (long long)rows * (cols / 4);
  long long blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(to_bf16_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)rows, cols, ld_src, ld_dst);
  return addhip::check_launch("to_bf16_kernel<x3>");
}

extern "C" int addhip_to_bf16_t(const float* src, uint16_t* dst, int32_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, void* stream) {
  ADDHIP_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "to_bf16_t: bad arguments");
  ADDHIP_RECORDABLE(addhip_to_bf16_t, src, dst, rows, cols, ld_src, ld_dst);
  hipLaunchKernelGGL(to_bf16_t_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols, ld_src, ld_dst);
  return addhip::check_launch("to_bf16_t_kernel");
}

extern "C" int addhip_shadow_refresh(const float* params, uint16_t* flat16, uint16_t* trans16, int64_t count, const int64_t* offset, const int32_t* rows,
                                     const int32_t* cols, int32_t n_mats, int32_t planes16, void* stream) {
  ADDHIP_REQUIRE(params && (flat16 || n_mats > 0) && count > 0 && n_mats >= 0 && n_mats <= ADDHIP_SHADOW_MAX_MATS, "shadow_refresh: bad arguments");
  ADDHIP_REQUIRE(n_mats == 0 || (trans16 && offset && rows && cols), "shadow_refresh: matrix table missing");
  ADDHIP_REQUIRE(aligned16(params) && (reinterpret_cast<uintptr_t>(flat16) & 7u) == 0, "shadow_refresh: misaligned buffers");
  const bool x3 = planes16 == ADDHIP_STORE_BF16X3;
  ADDHIP_REQUIRE(planes16 == 0 || planes16 == ADDHIP_STORE_BF16 || (x3 && count % 8 == 0 && aligned16(flat16) && aligned16(trans16)),
                 "shadow_refresh: planes16 is an ADDHIP_STORE_* format (plane storage: count %% 8 == 0, 16-byte aligned shadows)");
  ShadowMats mats;
  mats.n = n_mats > 0 ? n_mats : 1;
  int tiles = 0;
  for (int i = 0; i < ADDHIP_SHADOW_MAX_MATS; ++i) {
    const bool on = i < n_mats;
    ADDHIP_REQUIRE(!on || (offset[i] >= 0 && rows[i] > 0 && cols[i] > 0 && offset[i] + (int64_t)rows[i] * cols[i] <= count),
                   "shadow_refresh: a matrix lies outside the flat buffer");
    ADDHIP_REQUIRE(!on || !x3 || (offset[i] % 8 == 0 && rows[i] % 8 == 0), "shadow_refresh: plane storage needs matrix offsets and row counts %% 8 == 0");
    mats.offset[i] = on ? offset[i] : 0;
    mats.rows[i] = on ? rows[i] : 0;
    mats.cols[i] = on ? cols[i] : 0;
    if (on) tiles += ((rows[i] + 31) / 32) * ((cols[i] + 31) / 32);
    mats.tile_end[i] = tiles;
  }
  if (addhip::recording()) {  // (record.h: the host-side matrix table is copied)
    const std::vector<int64_t> o(offset, offset + n_mats);
    const std::vector<int32_t> r(rows, rows + n_mats), c(cols, cols + n_mats);
    return addhip::record_push(
        "addhip_shadow_refresh",
        [=](void* s) -> int { return addhip_shadow_refresh(params, flat16, trans16, count, o.data(), r.data(), c.data(), n_mats, planes16, s); }, nullptr, 0);
  }
  long long fb = (count / 4 + 255) / 256;
  const int flat_blocks = flat16 ? (int)(fb < 1 ? 1 : fb > 2048 ? 2048 : fb) : 0;  // flat16 == NULL: the transposed copies only
  if (x3)
    hipLaunchKernelGGL(shadow_refresh_kernel<true>, dim3((unsigned)(flat_blocks + tiles)), dim3(256), 0, (hipStream_t)stream, params, flat16, trans16,
                       (long long)count, flat_blocks, mats);
  else
    hipLaunchKernelGGL(shadow_refresh_kernel<false>, dim3((unsigned)(flat_blocks + tiles)), dim3(256), 0, (hipStream_t)stream, params, flat16, trans16,
                       (long long)count, flat_blocks, mats);
  return addhip::check_launch("shadow_refresh_kernel");
}

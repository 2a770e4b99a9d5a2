// fp32-class GEMM on operands STORED as three bf16 planes (ADDHIP_STORE_BF16X3, planes.h: x = hi + mid + lo exactly):
//   C[M,N] = alpha * sum_k A(m,k) * B(n,k),   a*b ~= hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid   (dropped: <= 2^-23 |a||b|)
// six v_mfma_f32_32x32x16_bf16 per k-step, fp32 accumulate: the error of the fp32 MFMA (gemm_split.hip states the bound) at 16/6 of its
// rate.  Unlike gemm_split.hip nothing is split here: the producers (GEMM epilogues, gather, loss heads, optimiser step) write the planes
// once, and the tiles travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).
//
// What bounds it, and the shape that follows.  Six MFMAs per staged fragment pair make this kernel MFMA-bound IF its tiles arrive: a
// 128x128 tile needs 31 bytes per clock and CU from L2 at the full matrix rate -- the whole of what the L2 -> LDS path delivers
// (~30 B/clk/CU measured) -- and a CU's 160 KB of LDS cannot hold enough 128x128 stages in flight to cover that path's latency under load
// (first version of this file: one 48 KB stage x 3 workgroups per CU, 165 TFLOP/s at 16384x1024x1024).  So the main configuration is a
// 256x256 tile (8 wavefronts of 128x64, one workgroup per CU): 15.6 B/clk/CU at the full matrix rate, a ring of three 16-deep K stages
// (48 KB each) with two stages in flight behind counted vmcnt waits and ONE raw barrier per stage (3072 MFMA cycles), no vmcnt(0)
// inside the loop.  256x128 (8 wavefronts of 64x64, ring of four) and 128x128 (4 wavefronts, ring of two, 3 workgroups per CU) cover the
// shapes that do not fill the chip with 256x256 tiles.
//
// Stage images (16 k of every row, three planes), either operand k-contiguous or m/n-contiguous in HBM:
//   k-contiguous   [row][96 B]: the row's two 48-byte k-groups, chunk (group g, plane p) at position (3g + p) ^ ((row >> 3) & 1) so that
//                  a ds_read_b128 lane group (16 rows at one chunk) covers all 64 banks; a lane's fragment = one 16-byte chunk;
//   m/n-contiguous [k-row][ROWS x 6 B]: groups of 8 rows x 3 planes, group ch at position ch ^ mc_swz(k-row); fragments by
//                  ds_read_b64_tr_b16 (no transposition on the way in).
// Both images are filled by "triples" of DMA instructions (192 lanes x 16 B = 3 KiB) that read WHOLE contiguous source runs (96 B of
// 32 rows / ROWS x 6 B of a k-row): the swizzle sits in the lane -> source mapping, the LDS side of a DMA is lane-linear.
//
// Replaces (agent.matmul_precision = bf16x3, update step): the torch.nn.Linear forward / backward of PPOModel / ADDModel
// (ppo_model.py:13-21, add_model.py:12-15) under fp32 semantics (the reference's CPU path; its GPU path is TF32, main.py:16-18).
#include "common.h"
#include "record.h"
#include "gemm_epilogue.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* ltr_t;

using addhip_epi::EPI_RUNTIME;
using addhip_epi::EpiBuf;
using addhip_epi::gemm_epilogue;

constexpr int BK = 16;  // k per stage: one MFMA k-step

__device__ uint4 g_zero_chunk3;  // 16 zero bytes: the DMA source of chunks beyond the end of K

__device__ __forceinline__ int mc_swz(int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); }

// tile configuration: WM x WN wavefronts, each (MT*32) x (NT*32); NS ring stages; BPC workgroups per CU
template <int WM_, int WN_, int MT_, int NT_, int NS_, int BPC_>
struct Cfg3 {
  static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_, NS = NS_, BPC = BPC_;
  static constexpr int NW = WM * WN, THREADS = 64 * NW;
  static constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  static constexpr int TILE_A = BM * 96, TILE_B = BN * 96, STAGE = TILE_A + TILE_B;  // bytes per stage (16 k x 6 B per row)
  static constexpr int TA = BM / 32, TB = BN / 32;                                      // DMA triples (3 KiB) per operand and stage
  static constexpr int TPW = (TA + TB + NW - 1) / NW;                                   // triples per wave (the last may be missing)
  static constexpr int EPI_BYTES = NW * EpiBuf<NT>::WAVE_BYTES;
  static constexpr int LDS_BYTES = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
  static constexpr int WPE = NW * BPC / 4;  // waves per SIMD (launch bounds)
};

// DMA addressing of this wave's triples: triple t of the stage (t < TA: operand A, else B) is wave t % NW's
template <typename Q, bool AKC, bool BKC>
struct Stager3 {
  unsigned off[Q::TPW][3];  // byte offset of this lane's chunk from the operand's (uniform) stage source
  int kq[Q::TPW][3];        // the k (relative to the stage's first k) the chunk starts at (KC) / lies on (MC): tail guard
  template <bool KC, int ROWS>
  __device__ __forceinline__ void init_one(int i, int t, int lane, int ld, int rows_left) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int L = 64 * s + lane;  // 0..191 within the triple
      if (KC) {  // 32 rows x 96 B
        const int row = 32 * t + L / 6, pos = L % 6, q = pos ^ ((row >> 3) & 1);
        off[i][s] = 6u * (unsigned)min(row, rows_left - 1) * (unsigned)ld + 16u * (unsigned)q;
        kq[i][s] = 8 * (q / 3);
      } else {   // 3072 bytes of the [k-row][ROWS x 6 B] image
        constexpr int CPR = ROWS * 6 / 16;  // chunks per k-row
        const int idx = 192 * t + L, kr = idx / CPR, pos = idx % CPR;
        const int cp = pos / 3, p = pos - 3 * cp, ch = cp ^ mc_swz(kr);
        off[i][s] = 6u * (unsigned)kr * (unsigned)ld + 48u * (unsigned)min(ch, rows_left / 8 - 1) + 16u * (unsigned)p;
        kq[i][s] = kr;
      }
    }
  }
  __device__ __forceinline__ void init(int wave, int lane, int lda, int ldb, int m_left, int n_left) {
#pragma unroll
    for (int i = 0; i < Q::TPW; ++i) {
      const int t = wave + i * Q::NW;  // (wave-uniform)
      if (t < Q::TA) init_one<AKC, Q::BM>(i, t, lane, lda, m_left);
      else init_one<BKC, Q::BN>(i, t - Q::TA, lane, ldb, n_left);
    }
  }
  // srcA / srcB: the operands at (tile's first row, stage's first k); slot: the stage's ring slot
  template <bool GUARD>
  __device__ __forceinline__ void issue(const char* srcA, const char* srcB, char* slot, int wave, int kleft) const {
#pragma unroll
    for (int i = 0; i < Q::TPW; ++i) {
      const int t = wave + i * Q::NW;
      if (t >= Q::TA + Q::TB) break;  // (only the last i of a wave can be missing)
      const char* src = t < Q::TA ? srcA : srcB;
      char* dst = slot + t * 3072;    // A's triples, then B's: TILE_A == TA * 3072
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const char* p = src + off[i][s];
        if (GUARD && kq[i][s] >= kleft) p = reinterpret_cast<const char*>(&g_zero_chunk3);  // K % 8 == 0 (KC): a chunk is in or out as a whole
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(dst + s * 1024), 16, 0, 0);
      }
    }
  }
};

// k-contiguous fragments: row w0 + 32a + li, plane p; lane half lh takes the second 8 k (k-group lh)
struct FragKC3 {
  unsigned base, x[3];
  __device__ __forceinline__ void init(int w0, int li, int lh) {  // w0 % 32 == 0
    base = (unsigned)(w0 + li) * 96u;
#pragma unroll
    for (int p = 0; p < 3; ++p) x[p] = (unsigned)((3 * lh + p) ^ ((li >> 3) & 1)) * 16u;
  }
  __device__ __forceinline__ bf16x8 get(const char* tile, int a, int p) const {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(tile + base + a * (32 * 96) + x[p]));
  }
};
// m/n-contiguous fragments: 8 consecutive k of one row per lane by two ds_read_b64_tr_b16 (each delivers a 4 k x 16 rows block transposed)
template <int ROWS, int MT>
struct FragMC3 {
  unsigned addr[MT][2];
  __device__ __forceinline__ void init(int w0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pl = i & 3;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kr = 8 * (g >> 1) + 4 * j + q;
        const int ch = (w0 + a * 32 + 16 * (g & 1)) / 8 + (pl >> 1);
        addr[a][j] = (unsigned)(ROWS * 6) * kr + 48u * (unsigned)(ch ^ mc_swz(kr)) + 8u * (pl & 1);
      }
  }
  __device__ __forceinline__ bf16x8 get(const char* tile, int a, int p) const {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(tile + addr[a][0] + 16 * p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(tile + addr[a][1] + 16 * p));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
};
template <bool KC, int ROWS, int MT> struct FragSel3 { typedef FragKC3 type; };
template <int ROWS, int MT> struct FragSel3<false, ROWS, MT> { typedef FragMC3<ROWS, MT> type; };

// s_waitcnt vmcnt(n) for a compile-time n
template <int N> __device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename Q, bool AKC, bool BKC, int EPI>
__global__ __launch_bounds__(Q::THREADS, Q::WPE) void gemm_x3_kernel(addhip_gemm_t g, int tiles_m, int tiles_n) {
  constexpr int BM = Q::BM, BN = Q::BN, MT = Q::MT, NT = Q::NT, NS = Q::NS;
  __shared__ __attribute__((aligned(1024))) char lds[Q::LDS_BYTES];

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

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm0 = (wave / Q::WN) * (MT * 32), wn0 = (wave % Q::WN) * (NT * 32);
  const int li = lane & 31, lh = lane >> 5;

  // stage sources: (tile's first row, first k of the split) and the step between stages; 6 bytes per value
  const char* srcA = reinterpret_cast<const char*>(g.A) + 6 * (AKC ? (size_t)m0 * g.lda + kbeg : (size_t)kbeg * g.lda + m0);
  const char* srcB = reinterpret_cast<const char*>(g.B) + 6 * (BKC ? (size_t)n0 * g.ldb + kbeg : (size_t)kbeg * g.ldb + n0);
  const size_t stepA = 6 * (AKC ? (size_t)BK : (size_t)BK * g.lda), stepB = 6 * (BKC ? (size_t)BK : (size_t)BK * g.ldb);
  Stager3<Q, AKC, BKC> sg;
  sg.init(wave, lane, g.lda, g.ldb, g.M - m0, g.N - n0);
  typename FragSel3<AKC, BM, MT>::type fa_;
  typename FragSel3<BKC, BN, NT>::type fb_;
  if constexpr (AKC) fa_.init(wm0, li, lh); else fa_.init(wm0, lane);
  if constexpr (BKC) fb_.init(wn0, li, lh); else fb_.init(wn0, lane);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;

  auto stage = [&](int kt) {  // kt < nk
    char* slot = lds + (kt % NS) * Q::STAGE;
    const int kleft = kend - (kbeg + kt * BK);
    if (kt < nk_full) sg.template issue<false>(srcA + kt * stepA, srcB + kt * stepB, slot, wave, kleft);
    else sg.template issue<true>(srcA + kt * stepA, srcB + kt * stepB, slot, wave, kleft);
  };
  // This wave's DMA instructions of one stage: 3 per triple it owns (waves past (TA + TB) % NW own one triple less).  The counted wait
  // below uses the smaller count PER LATER STAGE for every wave, i.e. it may wait for more than needed on waves that own more -- never less.
  constexpr int PIECES_MIN = 3 * ((Q::TA + Q::TB) / Q::NW);
  // Ring: stages kt+1 .. kt+NS-2 stay in flight while stage kt is multiplied.  Iteration kt: wait until this wave's share of stage kt
  // has landed (at most the later stages' pieces outstanding), barrier (every wave's share landed; every wave is done reading stage
  // kt-1, whose slot is the one stage kt+NS-1 goes to), issue stage kt+NS-1, multiply stage kt.
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s);
  for (int kt = 0; kt < nk; ++kt) {
    const int later = min(NS - 2, nk - 1 - kt);  // stages issued after kt that may still be in flight
    if (NS >= 4 && later >= 2) wait_vm<2 * PIECES_MIN>();
    else if (NS >= 3 && later >= 1) wait_vm<PIECES_MIN>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nk) stage(kt + NS - 1);
    const char* a_cur = lds + (kt % NS) * Q::STAGE;
    const char* b_cur = a_cur + Q::TILE_A;
    bf16x8 fa[MT][3], fb[NT][3];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int p = 0; p < 3; ++p) fb[b][p] = fb_.get(b_cur, b, p);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int p = 0; p < 3; ++p) fa[a][p] = fa_.get(a_cur, a, p);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) {  // smallest terms first
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
      }
  }

  __syncthreads();  // every wave is done with the last stage (and no DMA is in flight): LDS becomes the waves' private epilogue buffers
  gemm_epilogue<MT, NT, EPI, true>(g, acc, lds + wave * EpiBuf<NT>::WAVE_BYTES, lane, m0 + wm0, n0 + wn0, blockIdx.z, g.alpha);
}

typedef Cfg3<2, 4, 4, 2, 3, 1> CfgBig;   // 256x256, 8 waves of 128x64, ring of three, one workgroup per CU
typedef Cfg3<4, 2, 2, 2, 4, 1> CfgWide;  // 256x128, 8 waves of 64x64, ring of four, one workgroup per CU
typedef Cfg3<2, 2, 2, 2, 2, 3> CfgSmall; // 128x128, 4 waves of 64x64, ring of two, three workgroups per CU

template <typename Q>
int launch_x3(const addhip_gemm_t& g, hipStream_t st) {
  const int split = g.split_k > 1 ? g.split_k : 1;
  const int tiles_m = (g.M + Q::BM - 1) / Q::BM, tiles_n = (g.N + Q::BN - 1) / Q::BN;
  dim3 grid(tiles_m * tiles_n, 1, split), block(Q::THREADS);
#define ADDHIP_LAUNCH(AK, BKc, EPI) hipLaunchKernelGGL((gemm_x3_kernel<Q, AK, BKc, EPI>), grid, block, 0, st, g, tiles_m, tiles_n)
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
  return addhip::check_launch("gemm_x3_kernel");
}

}  // namespace

namespace addhip {
// called by addhip_gemm_f32 (gemm.hip) after argument validation, for descriptors whose operands are stored as bf16 planes
int gemm_x3_dispatch(const addhip_gemm_t& g, hipStream_t st) {
  if (g.a_mean || g.a_std) return (set_error("gemm: fused normalisation is not built for plane-stored operands"), -1);
  if (g.a_kcontig ? (g.K % 8 != 0 || g.lda % 8 != 0) : (g.M % 8 != 0 || g.lda % 8 != 0))
    return (set_error("gemm: plane-stored A needs whole 8-value groups (K or M, and lda, multiples of 8)"), -1);
  if (g.b_kcontig ? (g.K % 8 != 0 || g.ldb % 8 != 0) : (g.N % 8 != 0 || g.ldb % 8 != 0))
    return (set_error("gemm: plane-stored B needs whole 8-value groups (K or N, and ldb, multiples of 8)"), -1);
  if (!g.C && !g.C16) return (set_error("gemm: no output"), -1);
  if (g.split_k > 1 && (!g.C || g.C16)) return (set_error("gemm: split-K slabs are fp32"), -1);
  const long long split = g.split_k > 1 ? g.split_k : 1;
  auto tiles = [&](int bm, int bn) { return (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * split; };
  // the largest tile that still gives (nearly) every CU a workgroup; ADDHIP_GEMM_HINT_* override (tools, tests)
  int cfg = tiles(256, 256) >= 224 ? 0 : tiles(256, 128) >= 224 ? 1 : 2;
  if (g.hint & ADDHIP_GEMM_HINT_BIG_TILE) cfg = 0;
  if (g.hint & ADDHIP_GEMM_HINT_WIDE_TILE) cfg = 1;
  if (g.hint & ADDHIP_GEMM_HINT_NO_BIG_TILE) cfg = 2;
  return cfg == 0 ? launch_x3<CfgBig>(g, st) : cfg == 1 ? launch_x3<CfgWide>(g, st) : launch_x3<CfgSmall>(g, st);
}
}  // namespace addhip

// 128x128-tile MFMA GEMM whose operand tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write pass), shared by the fp32 path (gemm.hip: v_mfma_f32_32x32x2_f32) and the bf16-storage path (gemm_bf16.hip:
// v_mfma_f32_32x32x16_bf16):
//   C[M,N] = alpha * sum_k A(m,k) * B(n,k),  either operand k-contiguous or m/n-contiguous in HBM,
// epilogues / sign bits / column sums / split-K slabs by gemm_epilogue.h.
//
// One K stage is 128 BYTES of k per row for either element type (64 bf16 or 32 fp32), so the stage images, the DMA piece geometry
// and the k-contiguous fragment addressing are the same bytes for both; only the m/n-contiguous fragment reads differ
// (ds_read_b64_tr_b16 for bf16, four ds_read_b32 for fp32).  4 wavefronts of 64x64 (2x2 accumulators of 32x32); either two LDS
// stages (2 x 32 KB -> 2 workgroups per CU, the next stage's DMA in flight under this stage's MFMAs) or one 32 KB stage and 3-4
// workgroups per CU covering for each other.  The bank swizzle sits on the per-lane SOURCE address of the DMA (the LDS side of a
// DMA instruction is lane-linear: 64 x 16 B = 1 KiB at a wave-uniform address) and the same XOR is applied by the fragment reads.
//
// GROUPED launches: blockIdx.y selects one of up to ADDHIP_GEMM_MAX_GROUP equal-shaped problems (different buffers) -- the actor's
// and the critic's layers of an update step run as one launch, so that a launch has several rounds of tiles and one round's
// write-out overlaps the next round's first stage instead of the whole chip writing (and then fetching) in lock-step.
#pragma once
#include "common.h"
#include "gemm_epilogue.h"

namespace addhip_dma {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* ltr_t;

using addhip_epi::EPI_RUNTIME;
using addhip_epi::EpiBuf;
using addhip_epi::gemm_epilogue;

struct GemmGroup {
  addhip_gemm_t g[ADDHIP_GEMM_MAX_GROUP];
};

// 16 zero bytes: the DMA source of LDS chunks beyond the end of K (internal linkage: one copy per translation unit)
namespace { __device__ uint4 g_zero_chunk; }

// ---- element types.  CHE = elements per 16-byte chunk; BKS = k per stage (128 bytes); a fragment step consumes 2 chunks of k per
// row (lane half lh takes the second), i.e. 16 (bf16) / 8 (fp32) k, and there are 4 steps per stage.
struct ElemBF16 {
  static constexpr int ESZ = 2, CHE = 8, BKS = 64;
  typedef bf16x8 frag_t;
  // m/n-contiguous image: [64 k-rows][ROWS x 2 B]; chunk ch of k-row kr at position ch ^ mc_swz(kr) (16 chunks per k-row at ROWS = 128)
  static __device__ __forceinline__ int mc_swz(int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); }
  static __device__ __forceinline__ void mma(f32x16& acc, const frag_t& a, const frag_t& b) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); }
};
struct ElemF32 {
  static constexpr int ESZ = 4, CHE = 4, BKS = 32;
  typedef float4 frag_t;
  // m/n-contiguous image: [32 k-rows][ROWS x 4 B]; a ds_read_b32 of one k for rows li = 0..31 covers 32 banks, the other lane half reads
  // k + 4: its chunks are moved by 8 positions (32 banks) so that the 64 lanes cover all 64 banks
  static __device__ __forceinline__ int mc_swz(int kr) { return ((kr >> 2) & 1) << 3; }
  static __device__ __forceinline__ void mma(f32x16& acc, const frag_t& a, const frag_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  }
};

// k-contiguous operand P[r*ld + k] (activations, weights, transposed weight shadows): image [rows][128 B], 8 chunks per row, chunk c of
// row r at chunk position c ^ ((r >> 1) & 7); one DMA piece = 8 rows (a ds_read_b128 lane group = 16 rows at one k chunk covers all 64 banks)
__device__ __forceinline__ int kc_swz(int row) { return (row >> 1) & 7; }

// DMA addressing of one operand tile of ROWS rows: ROWS / 8 pieces of 1 KiB per stage, PPW per wave
template <typename E, bool KC, int ROWS, int PPW>
struct Stager {
  unsigned off[PPW];  // byte offset of this lane's 16-byte chunk from the stage's (uniform) source base, pieces PPW*wave .. PPW*wave+PPW-1
  int kq[PPW];        // the k (relative to the stage's first k) the chunk starts at (KC) / lies on (MC): tail guard
  // rows_left = R - r0 (>= 1; a multiple of CHE for MC)
  __device__ __forceinline__ void init(int wave, int lane, int ld, int rows_left) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = PPW * wave + i;
      if (KC) {
        const int row = 8 * piece + lane / 8, c = (lane % 8) ^ kc_swz(row);
        off[i] = (unsigned)E::ESZ * ((unsigned)min(row, rows_left - 1) * (unsigned)ld + (unsigned)(E::CHE * c));
        kq[i] = E::CHE * c;
      } else {
        constexpr int LPR = ROWS * E::ESZ / 16;  // lanes (16-byte chunks) per k-row of the image
        static_assert(64 % LPR == 0, "a DMA piece holds whole k-rows");
        const int kr = (64 / LPR) * piece + lane / LPR, ch = (lane % LPR) ^ E::mc_swz(kr);
        off[i] = (unsigned)E::ESZ * ((unsigned)kr * (unsigned)ld + (unsigned)min(E::CHE * ch, rows_left - E::CHE));
        kq[i] = kr;
      }
    }
  }
  // src: the operand at (tile's first row, stage's first k); dst: this operand's tile in the stage buffer; kleft = kend - k0
  template <bool GUARD>
  __device__ __forceinline__ void issue(const char* src, char* dst, int wave, int kleft) const {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const char* s = src + off[i];
      if (GUARD && kq[i] >= kleft) s = reinterpret_cast<const char*>(&g_zero_chunk);  // K % CHE == 0: a chunk is in or out as a whole
      __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(dst + (PPW * wave + i) * 1024), 16, 0, 0);
    }
  }
};

// fragment (rows row0 .. row0+31 of the tile, fragment step ks): lane (li, lh) gets row row0+li, chunk 2*ks + lh of its 128-byte row
template <typename E>
struct FragKC {
  unsigned base, x;  // row byte offset, lh ^ swizzle
  __device__ __forceinline__ void init(int w0, int li, int lh) { base = (unsigned)(w0 + li) * 128u; x = (unsigned)(lh ^ kc_swz(li)); }  // w0 % 32 == 0
  __device__ __forceinline__ typename E::frag_t get(const char* tile, int a, int ks) const {
    return __builtin_bit_cast(typename E::frag_t, *reinterpret_cast<const uint4*>(tile + base + a * (32 * 128) + (((2u * ks) ^ x) << 4)));
  }
};
template <typename E, int ROWS, int MT> struct FragMC;
// bf16: 8 consecutive k of one row per lane out of the [k][rows] image by two ds_read_b64_tr_b16 (each delivers a 4 k x 16 rows block transposed)
template <int ROWS, int MT>
struct FragMC<ElemBF16, ROWS, MT> {
  static constexpr unsigned RS = ROWS * 2;  // bytes per k-row of the image
  unsigned addr[MT][2];  // [a][j]: this lane's address for the j-th 4-k block of fragment a at ks = 0
  __device__ __forceinline__ void init(int w0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kr = 8 * (g >> 1) + 4 * j + q;                  // + 16*ks: leaves mc_swz(kr) unchanged
        const int ch = (w0 + a * 32 + 16 * (g & 1)) / 8 + (p >> 1);
        addr[a][j] = RS * kr + 16u * (ch ^ ElemBF16::mc_swz(kr)) + 8u * (p & 1);
      }
  }
  __device__ __forceinline__ bf16x8 get(const char* tile, int a, int ks) const {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(tile + addr[a][0] + ks * (16 * RS)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ltr_t)(tile + addr[a][1] + ks * (16 * RS)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
};
// fp32: k = 8*ks + 4*lh + j (j = 0..3) of row w0 + a*32 + li by four ds_read_b32; (kr >> 2) & 1 == lh for every one of them
template <int ROWS, int MT>
struct FragMC<ElemF32, ROWS, MT> {
  static constexpr unsigned RS = ROWS * 4;
  unsigned addr[MT];
  __device__ __forceinline__ void init(int w0, int lane) {
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int row = w0 + a * 32 + li;
      addr[a] = RS * (4u * lh) + 16u * ((unsigned)(row >> 2) ^ ((unsigned)lh << 3)) + 4u * (row & 3);
    }
  }
  __device__ __forceinline__ float4 get(const char* tile, int a, int ks) const {
    const char* p = tile + addr[a] + ks * (8 * RS);
    return make_float4(*reinterpret_cast<const float*>(p), *reinterpret_cast<const float*>(p + RS), *reinterpret_cast<const float*>(p + 2 * RS),
                       *reinterpret_cast<const float*>(p + 3 * RS));
  }
};
template <typename E, bool KC, int ROWS, int MT> struct FragSel { typedef FragKC<E> type; };
template <typename E, int ROWS, int MT> struct FragSel<E, false, ROWS, MT> { typedef FragMC<E, ROWS, MT> type; };

// tile configurations: CFG 0 = two LDS stages, 2 workgroups per CU; CFG 1 = one stage, 3-4 workgroups per CU
template <int CFG> struct Cfg {
  static constexpr int MT = 2, WN = 2, NW = 4;
  static constexpr int BM = (NW / WN) * MT * 32, BN = WN * 64;
  static constexpr int TILE_A = BM * 128, TILE_B = BN * 128, STAGE = TILE_A + TILE_B;
  static constexpr int STAGES = CFG == 1 ? 1 : 2;
  static constexpr int PPW = TILE_A / 1024 / NW;          // 1 KiB DMA pieces per wave, operand and stage (BM == BN)
  static constexpr int EPI_BYTES = NW * EpiBuf<2>::WAVE_BYTES;
  static constexpr int LDS_BYTES = STAGES * STAGE > EPI_BYTES ? STAGES * STAGE : EPI_BYTES;
};

// workgroups per CU: the one-stage configuration runs 4 (bf16 with a transposed operand: 3, its tr-read fragments need the registers)
template <typename E, bool AKC, bool BKC, int CFG> constexpr int min_blocks() { return CFG == 1 ? ((E::ESZ == 4 || (AKC && BKC)) ? 4 : 3) : 2; }

template <typename E, bool AKC, bool BKC, int EPI, int CFG>
__global__ __launch_bounds__(256, (min_blocks<E, AKC, BKC, CFG>())) void gemm_dma_kernel(GemmGroup grp, int tiles_m, int tiles_n) {
  typedef Cfg<CFG> Q;
  constexpr bool SB = Q::STAGES == 1;
  constexpr int BM = Q::BM, BN = Q::BN, MT = Q::MT, STAGE = Q::STAGE, TILE = Q::TILE_A;  // TILE: offset of the B tile in a stage
  constexpr int BK = E::BKS, ESZ = E::ESZ;
  __shared__ __attribute__((aligned(1024))) char lds[Q::LDS_BYTES];
  const addhip_gemm_t& g = grp.g[blockIdx.y];

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
  const int wm0 = (wave / Q::WN) * (MT * 32), wn0 = (wave % Q::WN) * 64;
  const int li = lane & 31, lh = lane >> 5;

  // stage sources: (tile's first row, first k of the split) and the step between stages
  const char* srcA = reinterpret_cast<const char*>(g.A) + (size_t)ESZ * (AKC ? (size_t)m0 * g.lda + kbeg : (size_t)kbeg * g.lda + m0);
  const char* srcB = reinterpret_cast<const char*>(g.B) + (size_t)ESZ * (BKC ? (size_t)n0 * g.ldb + kbeg : (size_t)kbeg * g.ldb + n0);
  const size_t stepA = (size_t)ESZ * (AKC ? (size_t)BK : (size_t)BK * g.lda), stepB = (size_t)ESZ * (BKC ? (size_t)BK : (size_t)BK * g.ldb);
  Stager<E, AKC, BM, Q::PPW> sa;
  Stager<E, BKC, BN, Q::PPW> sb;
  sa.init(wave, lane, g.lda, g.M - m0);
  sb.init(wave, lane, g.ldb, g.N - n0);
  typename FragSel<E, AKC, BM, MT>::type fa_;
  typename FragSel<E, BKC, BN, 2>::type fb_;
  if constexpr (AKC) fa_.init(wm0, li, lh); else fa_.init(wm0, lane);
  if constexpr (BKC) fb_.init(wn0, li, lh); else fb_.init(wn0, lane);

  f32x16 acc[MT][2];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;

  auto stage = [&](int kt, int buf) {
    char* dst = lds + buf * STAGE;
    const int kleft = kend - (kbeg + kt * BK);
    if (kt < nk_full) {
      sa.template issue<false>(srcA + kt * stepA, dst, wave, kleft);
      sb.template issue<false>(srcB + kt * stepB, dst + TILE, wave, kleft);
    } else {
      sa.template issue<true>(srcA + kt * stepA, dst, wave, kleft);
      sb.template issue<true>(srcB + kt * stepB, dst + TILE, wave, kleft);
    }
  };
  auto compute = [&](int buf) {
    const char* a_cur = lds + buf * STAGE;
    const char* b_cur = a_cur + TILE;
    // (the compiler sinks each step's fragment reads to just behind the issue of the MFMAs that consume the previous ones)
    typename E::frag_t fa[2][MT], fb[2][2];
#pragma unroll
    for (int a = 0; a < MT; ++a) fa[0][a] = fa_.get(a_cur, a, 0);
#pragma unroll
    for (int b = 0; b < 2; ++b) fb[0][b] = fb_.get(b_cur, b, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) {
#pragma unroll
        for (int a = 0; a < MT; ++a) fa[(ks + 1) & 1][a] = fa_.get(a_cur, a, ks + 1);
#pragma unroll
        for (int b = 0; b < 2; ++b) fb[(ks + 1) & 1][b] = fb_.get(b_cur, b, ks + 1);
      }
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) E::mma(acc[a][b], fa[ks & 1][a], fb[ks & 1][b]);
    }
  };
  {
    // Double-buffered (!SB): one barrier per stage.  Its vmcnt(0) retires this wave's share of stage kt (issued one whole
    // compute phase earlier), the barrier itself makes every wave's share visible and proves stage kt-1's buffer is no longer
    // being read, so the DMA of stage kt+1 into that buffer is issued right behind it and stays in flight under the MFMAs
    // of stage kt.  Single-buffered (SB): issue, wait + barrier, compute, barrier; the CU's other workgroups fill the waits.
    if (!SB && nk > 0) stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = SB ? 0 : (kt & 1);
      if (SB) {
        if (kt > 0) __syncthreads();
        stage(kt, 0);
      }
      __syncthreads();
      if (!SB && kt + 1 < nk) stage(kt + 1, cur ^ 1);
      compute(cur);
    }
  }

  __syncthreads();  // every wave is done with the last stage: LDS becomes the waves' private epilogue buffers
  gemm_epilogue<MT, 2, EPI>(g, acc, lds + wave * EpiBuf<2>::WAVE_BYTES, lane, m0 + wm0, n0 + wn0, blockIdx.z);
}

// hot layout / epilogue combinations get a compile-time epilogue; everything else shares the run-time one
template <typename E, int CFG>
inline void launch_dma(const GemmGroup& grp, int count, int tiles_m, int tiles_n, int split, hipStream_t st) {
  const addhip_gemm_t& g = grp.g[0];
  dim3 grid(tiles_m * tiles_n, count, split), block(256);
#define ADDHIP_DMA_LAUNCH(AK, BKc, EPI) hipLaunchKernelGGL((gemm_dma_kernel<E, AK, BKc, EPI, CFG>), grid, block, 0, st, grp, tiles_m, tiles_n)
  if (g.a_kcontig && g.b_kcontig) {
    if (g.epilogue == ADDHIP_EPI_BIAS_RELU) ADDHIP_DMA_LAUNCH(true, true, ADDHIP_EPI_BIAS_RELU);
    else if (g.epilogue == ADDHIP_EPI_MASK) ADDHIP_DMA_LAUNCH(true, true, ADDHIP_EPI_MASK);
    else ADDHIP_DMA_LAUNCH(true, true, EPI_RUNTIME);
  } else if (g.a_kcontig && !g.b_kcontig) {
    if (g.epilogue == ADDHIP_EPI_MASK) ADDHIP_DMA_LAUNCH(true, false, ADDHIP_EPI_MASK);
    else ADDHIP_DMA_LAUNCH(true, false, EPI_RUNTIME);
  } else if (!g.a_kcontig && g.b_kcontig) {
    ADDHIP_DMA_LAUNCH(false, true, EPI_RUNTIME);
  } else {
    if (g.epilogue == ADDHIP_EPI_NONE) ADDHIP_DMA_LAUNCH(false, false, ADDHIP_EPI_NONE);
    else ADDHIP_DMA_LAUNCH(false, false, EPI_RUNTIME);
  }
#undef ADDHIP_DMA_LAUNCH
}

}  // namespace addhip_dma

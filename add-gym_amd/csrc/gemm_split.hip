// fp32 GEMM through the bf16 matrix cores: every fp32 operand value is split EXACTLY into three bf16 chunks
// (x = hi + mid + lo, 8 significant bits each, by truncation: no rounding anywhere in the split) and the six products
// that carry bits above 2^-24 of |a||b| are accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//     a*b ~= hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid        (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |a||b|)
// A product of two bf16 values is exact in fp32, so the only roundings are those of the fp32 accumulation -- the same
// ones v_mfma_f32_32x32x2_f32 performs.  The CDNA4 bf16 MFMA rate is 16x its fp32 MFMA rate, so six bf16 MFMAs per
// k-step are 2.7x cheaper than the fp32 instruction for the same arithmetic (gemm.hip keeps the fp32-MFMA path).
// PLANES = 2 keeps hi and mid (16 significant bits per operand) and the three products above 2^-16: a TF32-class mode
// with 64x less error than TF32 at half the MFMA work of the exact split.  PLANES = 1 is the plain bf16 product
// (operands rounded toward zero to bf16, fp32 accumulate).
//
// F16 (ADDHIP_PREC_F16X2, include/addhip.h): two fp16 planes of x * s instead, s an exact power of two per operand tensor chosen from the
// tensor's tracked maximum (addhip_gemm_t.a_amax / b_amax) so that the largest value sits just below 2^15; hi = fp16(x s) and
// lo = fp16(x s - hi), round to nearest, keep 22 significant bits + the sign of the residual, and all FOUR products are formed
// (v_mfma_f32_32x32x16_f16): per-product error bound 2^-21 |a||b| at 4 instead of 6 matrix instructions per k-step and ~5 instead of ~9
// vector instructions per operand value.  The inverse of the two scales is folded into alpha (exact).
//
// Same interface, tiling and epilogues as gemm.hip: 128x128 tile, 4 wavefronts of 64x64 (2x2 MFMA accumulators),
// operands stay fp32 in HBM and are split on their way into LDS.  LDS image of an operand tile: [row][plane][16 k] bf16
// with a 16-byte pad per row (row stride 112 B = 28 dwords: a ds_read_b128 lane group covers all 64 banks); K advances 16
// per stage, LDS double-buffered, the next stage prefetched global -> VGPR while the current one feeds the MFMAs.
#include "common.h"
#include "gemm_epilogue.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 16;
using addhip_epi::EPI_RUNTIME;

template <int PLANES>
struct Img {
  static constexpr int RS = PLANES * 32 + 16;  // bytes per row
  static constexpr int SIZE = 128 * RS;        // bytes per operand tile
};

// x -> (hi, mid, lo) as fp32 bit patterns whose low 16 bits are zero (i.e. bf16 values), exact
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hi);
  mid = __float_as_uint(r1) & 0xffff0000u;
  lo = __float_as_uint(r1 - __uint_as_float(mid));  // <= 8 significant bits: already a bf16 value
}
// two bf16 (high halves of a, b) -> one dword, a in the low half
__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
// (x0, x1), already scaled -> dwords of their fp16 hi parts and fp16 lo parts (round to nearest; x - float(hi) is exact in fp32)
__device__ __forceinline__ void split2h(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f16x2 h = {(_Float16)x0, (_Float16)x1};
  const f16x2 l = {(_Float16)(x0 - (float)h[0]), (_Float16)(x1 - (float)h[1])};
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// power-of-two scale that puts `amax` (the maximum over the ADDHIP_AMAX_SLOTS float bit patterns at `slots`) into [2^14, 2^15): every
// thread of the workgroup computes it (64 L2-resident loads per wave); amax == 0 -> 1
__device__ __forceinline__ float scale_from_amax(const unsigned* __restrict__ slots) {
  unsigned m = slots[threadIdx.x & (ADDHIP_AMAX_SLOTS - 1)];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  const int e = (int)((m >> 23) & 0xffu);                  // biased exponent of amax: amax in [2^(e-127), 2^(e-126))
  const int se = min(max(268 - e, 1), 254);                // biased exponent of 2^(14 - (e - 127))
  return m == 0u ? 1.0f : __uint_as_float((unsigned)se << 23);
}

// ---- k-contiguous operand P[r*ld + k]: each thread moves 2 float4 (4 k of one row) per stage --------------------
// Loads are unconditional (addresses clamped into the K range) and NOTHING is done to the loaded registers until they
// are written to LDS several stages later (zeroing of the K tail, fused normalisation and the split all happen there):
// the K loop then has no branch around, and no early use of, a memory instruction, and the compiler keeps the whole
// register ring in flight across the loop back-edge (counted vmcnt).
__device__ __forceinline__ void load_kc(float4* reg, const float* __restrict__ P, int ld, int r0, int k0, int R, int kend) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = threadIdx.x + 256 * i;
    const int row = f >> 2, kq = (f & 3) * 4;
    const int r = min(r0 + row, R - 1), k = min(k0 + kq, kend - 4);
    reg[i] = *reinterpret_cast<const float4*>(P + (size_t)r * ld + k);
  }
}
// nmean / nstd: LDS copies of the normaliser vectors, indexed by k - kbeg (NORM only).  One call = one of the thread's two
// float4 (part i), so that the caller can spread the work between MFMA groups.  GUARD: K tail (k >= kend -> 0).
template <int PLANES, bool NORM, bool GUARD, bool F16 = false>
__device__ __forceinline__ void store_kc(char* lds, const float4* reg, int i, int k0, int kend, const float* nmean, const float* nstd, int kbeg, float sc = 1.f) {
  const int f = threadIdx.x + 256 * i;
  const int row = f >> 2, kq = (f & 3) * 4;
  const int k = k0 + kq;
  float4 v = reg[i];
  if (NORM) {  // Normalizer.normalize (normalizer.py:107-110)
    const int kc = min(k, kend - 4) - kbeg;
    const float4 mu = *reinterpret_cast<const float4*>(nmean + kc);
    const float4 sd = *reinterpret_cast<const float4*>(nstd + kc);
    v.x = (v.x - mu.x) / sd.x; v.y = (v.y - mu.y) / sd.y; v.z = (v.z - mu.z) / sd.z; v.w = (v.w - mu.w) / sd.w;
  }
  if (GUARD) {
    const bool in = k < kend;
    v = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
  }
  char* dst = lds + row * Img<PLANES>::RS + kq * 2;
  if (F16) {
    unsigned h0, l0, h1, l1;
    split2h(v.x * sc, v.y * sc, h0, l0);
    split2h(v.z * sc, v.w * sc, h1, l1);
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + 32) = make_uint2(l0, l1);
    return;
  }
  unsigned h[4], m[4], l[4];
  split3(v.x, h[0], m[0], l[0]);
  split3(v.y, h[1], m[1], l[1]);
  split3(v.z, h[2], m[2], l[2]);
  split3(v.w, h[3], m[3], l[3]);
  *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
  if (PLANES >= 2) *reinterpret_cast<uint2*>(dst + 32) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
  if (PLANES == 3) *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
}

// ---- m/n-contiguous operand P[k*ld + r]: each thread moves rows r..r+3 of two consecutive k per stage ---------------
__device__ __forceinline__ void load_mc(float4* reg, const float* __restrict__ P, int ld, int r0, int k0, int R, int kend) {
  const int kk2 = threadIdx.x & 7, rq = (threadIdx.x >> 3) * 4;
  const int r = min(r0 + rq, R - 4), k = k0 + 2 * kk2;
#pragma unroll
  for (int i = 0; i < 2; ++i) reg[i] = *reinterpret_cast<const float4*>(P + (size_t)min(k + i, kend - 1) * ld + r);
}
// part i = rows rq+2i, rq+2i+1
template <int PLANES, bool GUARD, bool F16 = false>
__device__ __forceinline__ void store_mc(char* lds, const float4* reg, int i, int k0, int kend, float sc = 1.f) {
  const int kk2 = threadIdx.x & 7, rq = (threadIdx.x >> 3) * 4;
  const int k = k0 + 2 * kk2;
  const bool in0 = !GUARD || k < kend, in1 = !GUARD || k + 1 < kend;
  const float a[2] = {i == 0 ? reg[0].x : reg[0].z, i == 0 ? reg[0].y : reg[0].w};  // k even
  const float b[2] = {i == 0 ? reg[1].x : reg[1].z, i == 0 ? reg[1].y : reg[1].w};  // k odd
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    char* dst = lds + (rq + 2 * i + j) * Img<PLANES>::RS + kk2 * 4;
    if (F16) {
      unsigned h, l;
      split2h(in0 ? a[j] * sc : 0.f, in1 ? b[j] * sc : 0.f, h, l);
      *reinterpret_cast<unsigned*>(dst) = h;
      *reinterpret_cast<unsigned*>(dst + 32) = l;
      continue;
    }
    unsigned ha, ma, la, hb, mb, lb;
    split3(in0 ? a[j] : 0.f, ha, ma, la);
    split3(in1 ? b[j] : 0.f, hb, mb, lb);
    *reinterpret_cast<unsigned*>(dst) = pack2(ha, hb);
    if (PLANES >= 2) *reinterpret_cast<unsigned*>(dst + 32) = pack2(ma, mb);
    if (PLANES == 3) *reinterpret_cast<unsigned*>(dst + 64) = pack2(la, lb);
  }
}

__device__ __forceinline__ bf16x8 frag(const char* lds, int row, int plane, int h, int rs) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lds + row * rs + plane * 32 + h * 16));
}

template <bool AKC, bool BKC, int EPI, bool NORM, int PLANES, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_split_kernel(addhip_gemm_t g, int tiles_m, int tiles_n) {
  static_assert(!NORM || AKC, "fused normalisation needs a k-contiguous A");
  static_assert(!F16 || (PLANES == 2 && !NORM), "the fp16 split keeps two planes and takes no fused normalisation (its scale needs the operand's maximum)");
  using I = Img<PLANES>;
  constexpr int STAGE = 2 * I::SIZE;
  constexpr int EPI_BYTES = 4 * addhip_epi::EpiBuf<2>::WAVE_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES];
  constexpr int NORM_MAXK = 512;  // fused normalisation: the first layer's K (= obs_stride)
  __shared__ __attribute__((aligned(16))) float nmean[NORM ? NORM_MAXK : 4], nstd[NORM ? NORM_MAXK : 4];

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

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;
  // F16: the operands' power-of-two scales; their inverse goes into the epilogue's alpha (exact)
  const float sc_a = F16 ? scale_from_amax(g.a_amax) : 1.f, sc_b = F16 ? scale_from_amax(g.b_amax) : 1.f;
  const float alpha = F16 ? g.alpha / sc_a / sc_b : g.alpha;

  // Register ring, DEPTH stages deep: one stage is only 24 MFMAs (768 cycles) per wave, far less than a global-load
  // round trip, so the loads of stage t+DEPTH are issued while stage t is multiplied.  Static ring indices (the K loop is
  // unrolled DEPTH stages per trip), no branch around a load, and a raw s_barrier with an LDS-only wait: __syncthreads()
  // would drain the ring (its fence waits for vmcnt(0)).  Inside a stage the split of the NEXT stage's registers (VALU)
  // is spread between the MFMA groups: an MFMA holds the SIMD's issue port for 8 of its 32 cycles, the rest is free.
  // Only whole 16-deep stages go through the pipeline; a K tail (K % 16) is one extra, guarded stage at the end.
  constexpr int DEPTH = 4;
  float4 ra[DEPTH][2], rb[DEPTH][2];
  const int nkf = kend > kbeg ? (kend - kbeg) / BK : 0;  // whole stages
  // (nk == 0: the K range of this split-K slice is empty -- more slices than 16-deep stages; its slab is all zeros, written
  //  by the epilogue below like any other)
  if (NORM && nk > 0) {
    for (int i = threadIdx.x; i < kend - kbeg; i += 256) { nmean[i] = g.a_mean[kbeg + i]; nstd[i] = g.a_std[kbeg + i]; }
    __syncthreads();
  }
  auto fetch = [&](int kt, float4* a_reg, float4* b_reg) {
    const int k0 = kbeg + min(kt, nk - 1) * BK;  // past the end: a redundant reload of the last stage, never consumed
    if (AKC) load_kc(a_reg, g.A, g.lda, m0, k0, g.M, kend);
    else load_mc(a_reg, g.A, g.lda, m0, k0, g.M, kend);
    if (BKC) load_kc(b_reg, g.B, g.ldb, n0, k0, g.N, kend);
    else load_mc(b_reg, g.B, g.ldb, n0, k0, g.N, kend);
  };
  // quarter q (0..3) of: registers of stage kt -> LDS buffer kt&1
  auto stash_part = [&](auto guard, int q, int kt, const float4* a_reg, const float4* b_reg) {
    constexpr bool GUARD = decltype(guard)::value;
    const int k0 = kbeg + min(kt, nk - 1) * BK;
    char* a_dst = lds + (kt & 1) * STAGE;
    if (q < 2) {
      if (AKC) store_kc<PLANES, NORM, GUARD, F16>(a_dst, a_reg, q, k0, kend, nmean, nstd, kbeg, sc_a);
      else store_mc<PLANES, GUARD, F16>(a_dst, a_reg, q, k0, kend, sc_a);
    } else {
      if (BKC) store_kc<PLANES, false, GUARD, F16>(a_dst + I::SIZE, b_reg, q - 2, k0, kend, nullptr, nullptr, 0, sc_b);
      else store_mc<PLANES, GUARD, F16>(a_dst + I::SIZE, b_reg, q - 2, k0, kend, sc_b);
    }
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto mfma_group = [&](int a, int b, const bf16x8 (*fa)[PLANES], const bf16x8 (*fb)[PLANES]) {
    if constexpr (F16) {  // all four products of the two-way fp16 split, smallest first (the fragments hold fp16 bit patterns)
      auto h = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[a][1]), h(fb[b][1]), acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[a][0]), h(fb[b][1]), acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[a][1]), h(fb[b][0]), acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[a][0]), h(fb[b][0]), acc[a][b], 0, 0, 0);
      return;
    }
    if (PLANES == 3) {  // smallest terms first
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][1], acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][2], acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][2], fb[b][0], acc[a][b], 0, 0, 0);
    }
    if (PLANES >= 2) {
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][1], acc[a][b], 0, 0, 0);
      acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][1], fb[b][0], acc[a][b], 0, 0, 0);
    }
    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a][0], fb[b][0], acc[a][b], 0, 0, 0);
  };
  auto read_frags = [&](int kt, bf16x8 (*fa)[PLANES], bf16x8 (*fb)[PLANES]) {
    const char* a_cur = lds + (kt & 1) * STAGE;
    const char* b_cur = a_cur + I::SIZE;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int p = 0; p < PLANES; ++p) fa[a][p] = frag(a_cur, wm0 + a * 32 + li, p, lh, I::RS);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int p = 0; p < PLANES; ++p) fb[b][p] = frag(b_cur, wn0 + b * 32 + li, p, lh, I::RS);
  };
  // one pipelined stage: slot s of the ring held stage kt (already in LDS buffer kt&1) and is refilled with stage
  // kt+DEPTH; stage kt+1 goes to the other LDS buffer (last read during stage kt-1, which every wave has left).  After
  // the last stage this writes a duplicate nobody reads.
  auto stage = [&](int kt, float4* a_free, float4* b_free, const float4* a_next, const float4* b_next) {
    fetch(kt + DEPTH, a_free, b_free);
    bf16x8 fa[2][PLANES], fb[2][PLANES];
    read_frags(kt, fa, fb);
    mfma_group(0, 0, fa, fb);
    stash_part(std::false_type{}, 0, kt + 1, a_next, b_next);
    mfma_group(0, 1, fa, fb);
    stash_part(std::false_type{}, 1, kt + 1, a_next, b_next);
    mfma_group(1, 0, fa, fb);
    stash_part(std::false_type{}, 2, kt + 1, a_next, b_next);
    mfma_group(1, 1, fa, fb);
    stash_part(std::false_type{}, 3, kt + 1, a_next, b_next);
    if (PLANES >= 2) {
      // pin the interleave: the split's VALU work spread evenly over the MFMAs, a DS write after every second (third) one
      // (F16: 4 / 6 / 8 vector instructions per MFMA and no pinning at all time within 1 % of each other: profiles/r04_split_pmc.json)
      constexpr int NM = F16 ? 16 : PLANES == 3 ? 24 : 12, PER = F16 ? 6 : PLANES == 3 ? 5 : 8;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, PER, 0);
        if (PLANES == 3 ? (i & 1) : true) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    lds_barrier();
  };
  if (nkf > 0) {  // (nkf > 0 implies nk > 0: the clamp min(kt, nk - 1) in fetch is then in range)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, ra[d], rb[d]);
#pragma unroll
    for (int q = 0; q < 4; ++q) stash_part(std::false_type{}, q, 0, ra[0], rb[0]);
    lds_barrier();
    int kt = 0;
    for (; kt + DEPTH <= nkf; kt += DEPTH) {
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) stage(kt + s, ra[s], rb[s], ra[(s + 1) % DEPTH], rb[(s + 1) % DEPTH]);
    }
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s)
      if (kt + s < nkf) stage(kt + s, ra[s], rb[s], ra[(s + 1) % DEPTH], rb[(s + 1) % DEPTH]);
  }
  if (nk > nkf) {  // K tail: one guarded, unpipelined stage (every wave is past the last barrier of the pipeline)
    fetch(nkf, ra[0], rb[0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) stash_part(std::true_type{}, q, nkf, ra[0], rb[0]);
    lds_barrier();
    bf16x8 fa[2][PLANES], fb[2][PLANES];
    read_frags(nkf, fa, fb);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) mfma_group(a, b, fa, fb);
  }

  // epilogue (gemm_epilogue.h): every wave's block leaves through its private slice of the stage buffers, once every wave is done
  // reading them
  __syncthreads();
  addhip_epi::gemm_epilogue<2, 2, EPI, false, true>(g, acc, lds + wave * addhip_epi::EpiBuf<2>::WAVE_BYTES, lane, m0 + wm0, n0 + wn0, blockIdx.z, alpha);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same kernel on 256-row tiles, 8 wavefronts (round 4).  Why: the split is vector-ALU work per STAGED element, the products are MFMA work
// per TILE element.  PMC of the 128x128 kernel (profiles/r04_split_pmc.json): matrix pipe busy 62 % (bf16x3), 47 % (f16x2), 43 % (bf16x2) --
// exactly what SIMD issue arithmetic predicts: a wave's stream needs 8 (MFMA) + 4 x (5..8 vector instructions) = 28..40 issue cycles per 32-cycle
// MFMA, and two waves share a SIMD.  A 256x256 tile stages the same 16 values per thread and stage but multiplies them twice as often
// (8 accumulators of 32x32 per wave instead of 4): 2.3-3.7 vector instructions per MFMA.  256x128 (8 waves of 64x64) for shapes that do not
// fill the chip with 256x256 tiles.  No fused normalisation here (those launches keep the 128x128 kernel).
template <int WM_, int WN_, int MT_, int NT_>
struct BigCfg {
  static constexpr int WM = WM_, WN = WN_, MT = MT_, NT = NT_;
  static constexpr int NW = WM * WN, THREADS = 64 * NW, BM = WM * MT * 32, BN = WN * NT * 32;
};

// k-contiguous operand of ROWS rows: LPT float4 (4 k of one row) per thread and stage
template <int THREADS, int ROWS>
struct KcMove {
  static constexpr int LPT = ROWS * 4 / THREADS;
  static_assert(LPT * THREADS == ROWS * 4 && LPT >= 1 && LPT <= 2, "whole float4 per thread");
  static __device__ __forceinline__ void load(float4* reg, const float* __restrict__ P, int ld, int r0, int k0, int R, int kend) {
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
      const int f = threadIdx.x + THREADS * i;
      const int row = f >> 2, kq = (f & 3) * 4;
      reg[i] = *reinterpret_cast<const float4*>(P + (size_t)min(r0 + row, R - 1) * ld + min(k0 + kq, kend - 4));
    }
  }
  template <int PLANES, bool GUARD, bool F16>
  static __device__ __forceinline__ void store(char* lds, const float4* reg, int i, int k0, int kend, float sc) {
    if (i >= LPT) return;
    const int f = threadIdx.x + THREADS * i;
    const int row = f >> 2, kq = (f & 3) * 4;
    float4 v = reg[i];
    if (GUARD) {
      const bool in = k0 + kq < kend;
      v = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
    }
    char* dst = lds + row * Img<PLANES>::RS + kq * 2;
    if (F16) {
      unsigned h0, l0, h1, l1;
      split2h(v.x * sc, v.y * sc, h0, l0);
      split2h(v.z * sc, v.w * sc, h1, l1);
      *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(dst + 32) = make_uint2(l0, l1);
      return;
    }
    unsigned h[4], m[4], l[4];
    split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]); split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
    if (PLANES >= 2) *reinterpret_cast<uint2*>(dst + 32) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
    if (PLANES == 3) *reinterpret_cast<uint2*>(dst + 64) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
  }
};
// m/n-contiguous operand of ROWS rows: thread (kk2, rq) moves rows rq..rq+3 of two consecutive k; THREADS / 2 rows are covered, threads
// beyond ROWS idle (wave-uniform: ROWS / 4 is a multiple of 64 threads' worth of rows)
template <int THREADS, int ROWS>
struct McMove {
  static_assert(THREADS / 2 >= ROWS, "the workgroup covers the tile's rows");
  static __device__ __forceinline__ bool mine() { return (int)(threadIdx.x >> 3) * 4 < ROWS; }
  static __device__ __forceinline__ void load(float4* reg, const float* __restrict__ P, int ld, int r0, int k0, int R, int kend) {
    const int kk2 = threadIdx.x & 7, rq = min((int)(threadIdx.x >> 3) * 4, ROWS - 4);
    const int r = min(r0 + rq, R - 4), k = k0 + 2 * kk2;
#pragma unroll
    for (int i = 0; i < 2; ++i) reg[i] = *reinterpret_cast<const float4*>(P + (size_t)min(k + i, kend - 1) * ld + r);
  }
  template <int PLANES, bool GUARD, bool F16>
  static __device__ __forceinline__ void store(char* lds, const float4* reg, int i, int k0, int kend, float sc) {
    if (!mine()) return;
    const int kk2 = threadIdx.x & 7, rq = (threadIdx.x >> 3) * 4;
    const int k = k0 + 2 * kk2;
    const bool in0 = !GUARD || k < kend, in1 = !GUARD || k + 1 < kend;
    const float a[2] = {i == 0 ? reg[0].x : reg[0].z, i == 0 ? reg[0].y : reg[0].w};  // k even
    const float b[2] = {i == 0 ? reg[1].x : reg[1].z, i == 0 ? reg[1].y : reg[1].w};  // k odd
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      char* dst = lds + (rq + 2 * i + j) * Img<PLANES>::RS + kk2 * 4;
      if (F16) {
        unsigned h, l;
        split2h(in0 ? a[j] * sc : 0.f, in1 ? b[j] * sc : 0.f, h, l);
        *reinterpret_cast<unsigned*>(dst) = h;
        *reinterpret_cast<unsigned*>(dst + 32) = l;
        continue;
      }
      unsigned ha, ma, la, hb, mb, lb;
      split3(in0 ? a[j] : 0.f, ha, ma, la);
      split3(in1 ? b[j] : 0.f, hb, mb, lb);
      *reinterpret_cast<unsigned*>(dst) = pack2(ha, hb);
      if (PLANES >= 2) *reinterpret_cast<unsigned*>(dst + 32) = pack2(ma, mb);
      if (PLANES == 3) *reinterpret_cast<unsigned*>(dst + 64) = pack2(la, lb);
    }
  }
};

template <typename Q, bool AKC, bool BKC, int EPI, int PLANES, bool F16>
__global__ __launch_bounds__(Q::THREADS, Q::NW / 4) void gemm_split_big_kernel(addhip_gemm_t g, int tiles_m, int tiles_n) {
  static_assert(PLANES >= 2 && (!F16 || PLANES == 2), "exact / two-chunk bf16 split, or the two-way fp16 split");
  using I = Img<PLANES>;
  constexpr int MT = Q::MT, NT = Q::NT, TH = Q::THREADS;
  constexpr int SIZE_A = Q::BM * I::RS, SIZE_B = Q::BN * I::RS, STAGE = SIZE_A + SIZE_B;
  constexpr int EPI_BYTES = Q::NW * addhip_epi::EpiBuf<NT>::WAVE_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES];
  typedef KcMove<TH, Q::BM> KA;
  typedef KcMove<TH, Q::BN> KB;
  typedef McMove<TH, Q::BM> MA;
  typedef McMove<TH, Q::BN> MB;
  constexpr int PA = AKC ? KA::LPT : 2, PB = BKC ? KB::LPT : 2;  // float4 registers (= stash parts) per operand and stage

  const int total = tiles_m * tiles_n;
  const int orig = blockIdx.x;
  const int q = total >> 3, r = total & 7, xcd = orig & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int tm = lin / tiles_n, tn = lin - tm * tiles_n;
  const int m0 = tm * Q::BM, n0 = tn * Q::BN;

  const int split = g.split_k > 1 ? g.split_k : 1;
  const int kchunk = ((g.K + split - 1) / split + BK - 1) / BK * BK;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(g.K, kbeg + kchunk);
  const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
  const int nkf = kend > kbeg ? (kend - kbeg) / BK : 0;  // whole stages

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave / Q::WN) * (MT * 32), wn0 = (wave % Q::WN) * (NT * 32);
  const int li = lane & 31, lh = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[a][b][x] = 0.f;
  const float sc_a = F16 ? scale_from_amax(g.a_amax) : 1.f, sc_b = F16 ? scale_from_amax(g.b_amax) : 1.f;
  const float alpha = F16 ? g.alpha / sc_a / sc_b : g.alpha;

  constexpr int DEPTH = 2;  // register ring: a stage is MT*NT*products MFMAs (1024-1536 cycles per wave, two waves per SIMD): two stages ahead cover a load round trip
  float4 ra[DEPTH][2], rb[DEPTH][2];
  auto fetch = [&](int kt, float4* a_reg, float4* b_reg) {
    const int k0 = kbeg + min(kt, nk - 1) * BK;  // past the end: a redundant reload of the last stage, never consumed
    if (AKC) KA::load(a_reg, g.A, g.lda, m0, k0, g.M, kend);
    else MA::load(a_reg, g.A, g.lda, m0, k0, g.M, kend);
    if (BKC) KB::load(b_reg, g.B, g.ldb, n0, k0, g.N, kend);
    else MB::load(b_reg, g.B, g.ldb, n0, k0, g.N, kend);
  };
  // part p (0 .. PA+PB-1) of: registers of stage kt -> LDS buffer kt & 1
  auto stash_part = [&](auto guard, int p, int kt, const float4* a_reg, const float4* b_reg) {
    constexpr bool GUARD = decltype(guard)::value;
    const int k0 = kbeg + min(kt, nk - 1) * BK;
    char* a_dst = lds + (kt & 1) * STAGE;
    if (p < PA) {
      if (AKC) KA::template store<PLANES, GUARD, F16>(a_dst, a_reg, p, k0, kend, sc_a);
      else MA::template store<PLANES, GUARD, F16>(a_dst, a_reg, p, k0, kend, sc_a);
    } else {
      if (BKC) KB::template store<PLANES, GUARD, F16>(a_dst + SIZE_A, b_reg, p - PA, k0, kend, sc_b);
      else MB::template store<PLANES, GUARD, F16>(a_dst + SIZE_A, b_reg, p - PA, k0, kend, sc_b);
    }
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto products = [&](f32x16& c, const bf16x8* fa, const bf16x8* fb) {
    if constexpr (F16) {  // all four products of the two-way fp16 split, smallest first (the fragments hold fp16 bit patterns)
      auto h = [](const bf16x8& v) { return __builtin_bit_cast(f16x8, v); };
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[1]), h(fb[1]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[0]), h(fb[1]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[1]), h(fb[0]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(h(fa[0]), h(fb[0]), c, 0, 0, 0);
    } else {
      if (PLANES == 3) {  // smallest terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2 % PLANES], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 % PLANES], fb[0], c, 0, 0, 0);
      }
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], c, 0, 0, 0);
    }
  };
  // the MFMAs of stage kt (LDS buffer kt & 1) with the stash of stage kt + 1 spread between the accumulator rows; B fragments for the
  // whole stage, A fragments one 32-row block at a time (registers)
  auto multiply = [&](auto guard, int kt, bool stash, const float4* a_next, const float4* b_next) {
    const char* a_cur = lds + (kt & 1) * STAGE;
    const char* b_cur = a_cur + SIZE_A;
    bf16x8 fb[NT][PLANES];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int p = 0; p < PLANES; ++p) fb[b][p] = frag(b_cur, wn0 + b * 32 + li, p, lh, I::RS);
    int done = 0;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      bf16x8 fa[PLANES];
#pragma unroll
      for (int p = 0; p < PLANES; ++p) fa[p] = frag(a_cur, wm0 + a * 32 + li, p, lh, I::RS);
#pragma unroll
      for (int b = 0; b < NT; ++b) products(acc[a][b], fa, fb[b]);
      const int upto = (a + 1) * (PA + PB) / MT;  // parts stashed once this accumulator row is issued
      if (stash)
        for (; done < upto; ++done) stash_part(guard, done, kt + 1, a_next, b_next);
    }
  };
  if (nkf > 0) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch(d, ra[d], rb[d]);
#pragma unroll
    for (int p = 0; p < PA + PB; ++p) stash_part(std::false_type{}, p, 0, ra[0], rb[0]);
    lds_barrier();
    int kt = 0;
    for (; kt + DEPTH <= nkf; kt += DEPTH) {
#pragma unroll
      for (int s = 0; s < DEPTH; ++s) {
        // slot s held stage kt+s (already in LDS); it is refilled with stage kt+s+DEPTH while stage kt+s is multiplied and stage kt+s+1
        // (slot (s+1) % DEPTH) goes to the other LDS buffer
        fetch(kt + s + DEPTH, ra[s], rb[s]);
        multiply(std::false_type{}, kt + s, true, ra[(s + 1) % DEPTH], rb[(s + 1) % DEPTH]);
        lds_barrier();
      }
    }
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s)
      if (kt + s < nkf) {
        fetch(kt + s + DEPTH, ra[s], rb[s]);
        multiply(std::false_type{}, kt + s, true, ra[(s + 1) % DEPTH], rb[(s + 1) % DEPTH]);
        lds_barrier();
      }
  }
  if (nk > nkf) {  // K tail: one guarded, unpipelined stage (every wave is past the last barrier of the pipeline)
    fetch(nkf, ra[0], rb[0]);
#pragma unroll
    for (int p = 0; p < PA + PB; ++p) stash_part(std::true_type{}, p, nkf, ra[0], rb[0]);
    lds_barrier();
    multiply(std::true_type{}, nkf, false, ra[0], rb[0]);
  }
  __syncthreads();
  addhip_epi::gemm_epilogue<MT, NT, EPI, false, true>(g, acc, lds + wave * addhip_epi::EpiBuf<NT>::WAVE_BYTES, lane, m0 + wm0, n0 + wn0, blockIdx.z, alpha);
}

typedef BigCfg<2, 4, 4, 2> SplitBig;   // 256x256: 8 waves of 128x64
typedef BigCfg<4, 2, 2, 2> SplitWide;  // 256x128: 8 waves of 64x64

template <typename Q, int PLANES, bool F16>
int launch_split_big(const addhip_gemm_t& g, hipStream_t st) {
  const int tiles_m = (g.M + Q::BM - 1) / Q::BM, tiles_n = (g.N + Q::BN - 1) / Q::BN;
  const int split = g.split_k > 1 ? g.split_k : 1;
  dim3 grid(tiles_m * tiles_n, 1, split), block(Q::THREADS);
#define ADDHIP_LAUNCH(AK, BKc, EPI) hipLaunchKernelGGL((gemm_split_big_kernel<Q, AK, BKc, EPI, PLANES, F16>), grid, block, 0, st, g, tiles_m, tiles_n)
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
  return addhip::check_launch("gemm_split_big_kernel");
}

template <int PLANES, bool F16 = false>
int launch_split(const addhip_gemm_t& g, hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
  const int split = g.split_k > 1 ? g.split_k : 1;
  dim3 grid(tiles_m * tiles_n, 1, split), block(256);
  const bool norm = g.a_mean != nullptr;
#define ADDHIP_LAUNCH(AK, BKc, EPI, NORM) \
  hipLaunchKernelGGL((gemm_split_kernel<AK, BKc, EPI, (NORM) && !F16, PLANES, F16>), grid, block, 0, st, g, tiles_m, tiles_n)
  if (g.a_kcontig && g.b_kcontig) {
    if (norm) {
      if (F16) return (addhip::set_error("gemm: the fp16 split takes no fused normalisation"), -1);
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
  return addhip::check_launch("gemm_split_kernel");
}

}  // namespace

namespace addhip {
// called by addhip_gemm_f32 (gemm.hip) after argument validation, for the shapes that fill the chip with 128x128 tiles
int gemm_split_dispatch(const addhip_gemm_t& g, int planes, hipStream_t st) {  // planes: ADDHIP_PREC_* (1..4)
  if (g.a_mean && g.K > 512) return (set_error("gemm: fused normalisation on the bf16 paths needs K <= 512"), -1);
  // ADDHIP_PREC_F16X2 needs both operand bounds and no fused normalisation (the normalised values' maximum is not tracked): otherwise the
  // exact bf16 split, which needs neither
  const bool f16 = planes == ADDHIP_PREC_F16X2 && g.a_amax && g.b_amax && !g.a_mean;
  if (planes == ADDHIP_PREC_F16X2 && !f16) planes = ADDHIP_PREC_BF16X3;
  // 256-row tiles (8 waves: half the split arithmetic per MFMA) where they fill the chip; ADDHIP_GEMM_HINT_* override (tools, tests)
  const long long split = g.split_k > 1 ? g.split_k : 1;
  auto wgs = [&](int bm, int bn) { return (long long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn) * split; };
  int cfg = g.a_mean || planes == ADDHIP_PREC_BF16 ? 2 : wgs(256, 256) >= 224 ? 0 : wgs(256, 128) >= 224 ? 1 : 2;
  // (fp16 split, ragged last N tile on 256x128 tiles: the first-layer weight gradients, N = 272 -- 93 us there against 77 on 128x128 tiles)
  if (f16 && cfg == 1 && g.N % 128 != 0) cfg = 2;
  if (!g.a_mean && planes != ADDHIP_PREC_BF16) {
    if (g.hint & ADDHIP_GEMM_HINT_BIG_TILE) cfg = 0;
    if (g.hint & ADDHIP_GEMM_HINT_WIDE_TILE) cfg = 1;
    if (g.hint & ADDHIP_GEMM_HINT_NO_BIG_TILE) cfg = 2;
  }
  if (cfg == 0) return f16 ? launch_split_big<SplitBig, 2, true>(g, st) : planes == 3 ? launch_split_big<SplitBig, 3, false>(g, st) : launch_split_big<SplitBig, 2, false>(g, st);
  if (cfg == 1) return f16 ? launch_split_big<SplitWide, 2, true>(g, st) : planes == 3 ? launch_split_big<SplitWide, 3, false>(g, st) : launch_split_big<SplitWide, 2, false>(g, st);
  if (f16) return launch_split<2, true>(g, st);
  return planes == 3 ? launch_split<3>(g, st) : planes == 2 ? launch_split<2>(g, st) : launch_split<1>(g, st);
}
}  // namespace addhip

// Shared epilogue of the MFMA GEMM kernels (gemm.hip, gemm_split.hip, gemm_bf16.hip): one implementation of the bias / ReLU /
// mask epilogues, the ReLU sign bits, the fused bias-gradient column sums and the write-out of C (fp32, optionally added by
// atomics) and C16 (bf16, round to nearest even).
#pragma once
#include "common.h"
#include "planes.h"

namespace addhip_epi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int EPI_RUNTIME = -1;  // epilogue chosen from the descriptor at run time (cold combinations)

// fp32 -> bf16, round to nearest even (no NaN special-casing: the callers' values are finite)
__device__ __forceinline__ unsigned short epi_bf16(float v) {
  const unsigned u = __float_as_uint(v);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Epilogue of a wave's (MT*32) x (NT*32) accumulator block (v_mfma_f32_32x32x* layout) whose first element is (row0, col0) of C.  In the accumulators a lane owns
// ONE column (col0+b*32+li) and register x is row (x&3)+8*(x>>2)+4*lh of the 32x32 tile: bias, ReLU, mask, sign bits and the
// bias-gradient column sums are done in that layout, then each 32-row slice goes through a wave-private LDS buffer and leaves as
// 16-byte stores along the rows (2-byte and 4-byte stores straight from the accumulators cost 4-8x the store instructions,
// which is what a short-K launch then spends its time on).  The caller has passed a barrier behind the last LDS stage.
template <int NT> struct EpiBuf { static constexpr int ERS = NT * 128 + 16, WAVE_BYTES = 32 * ERS; };  // bytes per staged row (+ one 16-byte pad), per wave

// offset of the colsum row a 32-row block adds to (addhip_gemm_t.colsum_replicas: same-line float atomics of all row tiles serialise)
__device__ __forceinline__ size_t epi_cs_row(const addhip_gemm_t& g, int rtile) {
  return g.colsum_replicas > 1 ? (size_t)((rtile >> 5) % g.colsum_replicas) * (size_t)g.ldcs : (size_t)0;
}

typedef __bf16 epi_bf16x2 __attribute__((ext_vector_type(2)));
typedef float epi_f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> one dword of two bf16 (round to nearest even) by the hardware conversion (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned epi_pack_bf16(float lo, float hi) {
  const epi_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, epi_bf16x2));
}

// r[lane] = v (v_writelane_b32) through the compiler's own intrinsic, so that it schedules the wait states a v_writelane needs behind the
// v_cmp that produced its scalar operand (hand-written inline asm gets none and reads a stale mask)
extern "C" __device__ unsigned addhip_llvm_writelane(unsigned val, unsigned lane, unsigned old) __asm("llvm.amdgcn.writelane.i32");

// The wave's block lies wholly inside C and every output row is 16-byte aligned (all the tiles of this path's layer shapes but a few
// edge ones): the same epilogue as straight-line code -- no per-element bounds, no alignment fallbacks, no atomics.  The general form
// below executes ~2000 instructions per wave, which with 16 waves per CU is what a short-K launch took its time for (10-17 us); this
// one ~600.  Masking is an AND with the sign-extended bit of the lane's column; bf16 results are rounded by v_cvt_pk_bf16_f32.
// max |v| of a wave's block -> one atomic per wave on slot (workgroup % ADDHIP_AMAX_SLOTS) (addhip_gemm_t.amax_out; non-negative floats order as
// their bit patterns)
__device__ __forceinline__ void epi_amax(const addhip_gemm_t& g, float amx) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(&g.amax_out[(blockIdx.x + blockIdx.y * gridDim.x) % ADDHIP_AMAX_SLOTS], __float_as_uint(amx));
}

// X3OUT: the kernel may be asked for a plane-storage C16 (addhip_gemm_t.c16_planes = ADDHIP_STORE_BF16X3).  Only gemm_x3.hip instantiates it:
// the split of 8 values into three planes needs ~30 more registers, which the 4-workgroups-per-CU kernels (128 VGPRs) do not have.
// AMAX: the kernel honours addhip_gemm_t.amax_out (the register-staged and split kernels: gemm.hip's gemm_kernel, gemm_split.hip; the
// dispatcher sends descriptors that carry amax_out to those).
template <int MT, int NT, int EPI, bool X3OUT = false, bool AMAX = false>
__device__ __forceinline__ void gemm_epilogue_full(const addhip_gemm_t& g, f32x16 (&acc)[MT][NT], char* ebuf, int lane, int row0, int col0, float* C,
                                                   unsigned short* C16, float alpha) {
  constexpr int ERS = EpiBuf<NT>::ERS;
  const int li = lane & 31, lh = lane >> 5;
  const int epi = EPI == EPI_RUNTIME ? g.epilogue : EPI;
  const bool has_bias = epi == ADDHIP_EPI_BIAS || epi == ADDHIP_EPI_BIAS_RELU;
  const bool want_bits = epi == ADDHIP_EPI_BIAS_RELU && g.relu_bits != nullptr;
  const bool want_cs = epi == ADDHIP_EPI_MASK && g.colsum != nullptr;
  float amx = 0.f;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int rtile = row0 + a * 32;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int cgroup = col0 + b * 32, col = cgroup + li;
      const float bias = has_bias ? g.bias[col] : 0.f;
      // sign-bit word of tile row `lane` (lanes 0..31), fetched once and handed out by readlane (this form is only taken with mask_bits)
      unsigned mword = 0u;
      if (epi == ADDHIP_EPI_MASK && lane < 32) mword = g.mask_bits[(size_t)(rtile + lane) * g.ldbits + (cgroup >> 5)];
      unsigned rword = 0u;
      float cs = 0.f;
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int r0 = (x & 3) + 8 * (x >> 2), rloc = r0 + 4 * lh;
        float v = alpha * acc[a][b][x] + bias;
        if (epi == ADDHIP_EPI_BIAS_RELU) {
          v = fmaxf(v, 0.f);
          if (want_bits) {
            const unsigned long long pos = __ballot(v > 0.f);
            // rows r0 (lanes 0-31 of the ballot) and r0 + 4 (lanes 32-63): each half goes into the lane that owns that row's word
            rword = addhip_llvm_writelane((unsigned)pos, r0, rword);
            rword = addhip_llvm_writelane((unsigned)(pos >> 32), r0 + 4, rword);
          }
        }
        if (epi == ADDHIP_EPI_MASK) {
          const unsigned w0 = __builtin_amdgcn_readlane(mword, r0), w1 = __builtin_amdgcn_readlane(mword, r0 + 4);
          const int keep = __builtin_amdgcn_sbfe((int)(lh ? w1 : w0), li, 1);  // 0 or -1: the sign-extended bit of this lane's column
          v = __uint_as_float(__float_as_uint(v) & (unsigned)keep);
          cs += v;
        }
        if (AMAX) amx = fmaxf(amx, fabsf(v));
        *reinterpret_cast<float*>(ebuf + rloc * ERS + (b * 32 + li) * 4) = v;
      }
      if (want_bits && lane < 32) g.relu_bits[(size_t)(rtile + lane) * g.ldbits + (cgroup >> 5)] = rword;
      if (want_cs) {
        cs += __shfl_xor(cs, 32, 64);
        if (lh == 0) atomicAdd(&g.colsum[epi_cs_row(g, rtile) + col], cs);
      }
    }
    if (C) {  // NT*8 lanes x 4 columns per row
      constexpr int CPR = NT * 8;
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        const int idx = lane + 64 * i, rloc = idx / CPR, c4 = (idx % CPR) * 4;
        *reinterpret_cast<float4*>(C + (size_t)(rtile + rloc) * g.ldc + col0 + c4) = *reinterpret_cast<const float4*>(ebuf + rloc * ERS + c4 * 4);
      }
    }
    if (C16) {  // NT*4 lanes x 8 columns per row
      constexpr int CPR = NT * 4;
#pragma unroll
      for (int i = 0; i < NT * 2; ++i) {
        const int idx = lane + 64 * i, rloc = idx / CPR, c8 = (idx % CPR) * 8;
        const float4 lo = *reinterpret_cast<const float4*>(ebuf + rloc * ERS + c8 * 4);
        const float4 hi = *reinterpret_cast<const float4*>(ebuf + rloc * ERS + c8 * 4 + 16);
        if (X3OUT && g.c16_planes == ADDHIP_STORE_BF16X3)  // plane storage: the exact 3-way split, 48 contiguous bytes per 8 columns
          addhip_planes::store8(C16 + 3 * (size_t)(rtile + rloc) * g.ldc16, col0 + c8, lo, hi);
        else
          *reinterpret_cast<uint4*>(C16 + (size_t)(rtile + rloc) * g.ldc16 + col0 + c8) =
              make_uint4(epi_pack_bf16(lo.x, lo.y), epi_pack_bf16(lo.z, lo.w), epi_pack_bf16(hi.x, hi.y), epi_pack_bf16(hi.z, hi.w));
      }
    }
  }
  if (AMAX && g.amax_out) epi_amax(g, amx);
}

// alpha: the factor applied to the accumulators (g.alpha; the F16X2 kernel folds the inverse of its operand scales in)
template <int MT, int NT, int EPI, bool X3OUT = false, bool AMAX = false>
__device__ __forceinline__ void gemm_epilogue(const addhip_gemm_t& g, f32x16 (&acc)[MT][NT], char* ebuf, int lane, int row0, int col0, int zslab, float alpha) {
  constexpr int ERS = EpiBuf<NT>::ERS;
  typedef unsigned short u16;
  const int li = lane & 31, lh = lane >> 5;
  const int epi = EPI == EPI_RUNTIME ? g.epilogue : EPI;
  const bool accum = g.accumulate != 0;  // K slices add into one C (hardware fp32 atomics) instead of writing slabs
  float* C = g.C ? g.C + (accum ? (size_t)0 : (size_t)zslab * (size_t)g.M * g.ldc) : nullptr;  // zslab: the split-K slice this block computed
  u16* C16 = reinterpret_cast<u16*>(g.C16);
  const bool c_vec = C && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && (g.ldc & 3) == 0;
  const bool c16_vec = C16 && (reinterpret_cast<uintptr_t>(C16) & 15) == 0 && (g.ldc16 & 7) == 0;
  // (wave-uniform) the straight-line form for blocks wholly inside C with aligned rows
  if (row0 + MT * 32 <= g.M && col0 + NT * 32 <= g.N && !accum && (!C || c_vec) && (!C16 || c16_vec) && (epi != ADDHIP_EPI_MASK || g.mask_bits)) {
    gemm_epilogue_full<MT, NT, EPI, X3OUT, AMAX>(g, acc, ebuf, lane, row0, col0, C, C16, alpha);
    return;
  }
  float amx = 0.f;
#pragma unroll
  for (int a = 0; a < MT; ++a) {
    const int rtile = row0 + a * 32;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int cgroup = col0 + b * 32, col = cgroup + li;
      const bool col_ok = col < g.N;
      const float bias = (col_ok && (epi == ADDHIP_EPI_BIAS || epi == ADDHIP_EPI_BIAS_RELU)) ? g.bias[col] : 0.f;
      // sign-bit word of tile row `lane` (lanes 0..31), fetched once and handed out by readlane
      unsigned mword = 0u;
      if (epi == ADDHIP_EPI_MASK && g.mask_bits && lane < 32 && cgroup < g.N && rtile + lane < g.M)
        mword = g.mask_bits[(size_t)(rtile + lane) * g.ldbits + (cgroup >> 5)];
      unsigned rword = 0u;  // lanes 0..31: the ReLU sign-bit word of tile row `lane`
      float cs = 0.f;
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int r0 = (x & 3) + 8 * (x >> 2), rloc = r0 + 4 * lh, row = rtile + rloc;
        const bool ok = col_ok && row < g.M;
        float v = alpha * acc[a][b][x] + bias;
        if (epi == ADDHIP_EPI_BIAS_RELU) v = fmaxf(v, 0.f);
        if (epi == ADDHIP_EPI_MASK) {
          if (g.mask_bits) {
            const unsigned w0 = __builtin_amdgcn_readlane(mword, r0), w1 = __builtin_amdgcn_readlane(mword, r0 + 4);
            v = (((lh ? w1 : w0) >> li) & 1u) ? v : 0.f;
          } else {
            v = (ok && g.mask[(size_t)row * g.ldmask + col] > 0.f) ? v : 0.f;
          }
          if (ok) cs += v;
        }
        if (AMAX && ok) amx = fmaxf(amx, fabsf(v));
        if (epi == ADDHIP_EPI_BIAS_RELU && g.relu_bits) {
          const unsigned long long pos = __ballot(ok && v > 0.f);
          rword = lane == r0 ? (unsigned)pos : lane == r0 + 4 ? (unsigned)(pos >> 32) : rword;
        }
        *reinterpret_cast<float*>(ebuf + rloc * ERS + (b * 32 + li) * 4) = v;
      }
      if (epi == ADDHIP_EPI_BIAS_RELU && g.relu_bits && lane < 32 && cgroup < g.N && rtile + lane < g.M)
        g.relu_bits[(size_t)(rtile + lane) * g.ldbits + (cgroup >> 5)] = rword;
      if (epi == ADDHIP_EPI_MASK && g.colsum) {
        cs += __shfl_xor(cs, 32, 64);
        if (lh == 0 && col_ok) atomicAdd(&g.colsum[epi_cs_row(g, rtile) + col], cs);
      }
    }
    if (C) {  // NT*8 lanes x 4 columns per row
      constexpr int CPR = NT * 8;
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        const int idx = lane + 64 * i, rloc = idx / CPR, c4 = (idx % CPR) * 4, row = rtile + rloc, col = col0 + c4;
        const float4 v = *reinterpret_cast<const float4*>(ebuf + rloc * ERS + c4 * 4);
        if (row < g.M) {
          float* dst = C + (size_t)row * g.ldc + col;
          if (c_vec && !accum && col + 3 < g.N) {
            *reinterpret_cast<float4*>(dst) = v;
          } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (col + j < g.N) {
                if (accum) unsafeAtomicAdd(&dst[j], e[j]);
                else dst[j] = e[j];
              }
          }
        }
      }
    }
    if (C16) {  // NT*4 lanes x 8 columns per row
      constexpr int CPR = NT * 4;
#pragma unroll
      for (int i = 0; i < NT * 2; ++i) {
        const int idx = lane + 64 * i, rloc = idx / CPR, c8 = (idx % CPR) * 8, row = rtile + rloc, col = col0 + c8;
        const float4 lo = *reinterpret_cast<const float4*>(ebuf + rloc * ERS + c8 * 4);
        const float4 hi = *reinterpret_cast<const float4*>(ebuf + rloc * ERS + c8 * 4 + 16);
        if (X3OUT && row < g.M && g.c16_planes == ADDHIP_STORE_BF16X3) {  // (N % 8 == 0: a group of 8 columns is in or out as a whole)
          if (col < g.N) addhip_planes::store8(C16 + 3 * (size_t)row * g.ldc16, col, lo, hi);
        } else if (row < g.M) {
          u16* dst = C16 + (size_t)row * g.ldc16 + col;
          const u16 e[8] = {epi_bf16(lo.x), epi_bf16(lo.y), epi_bf16(lo.z), epi_bf16(lo.w), epi_bf16(hi.x), epi_bf16(hi.y), epi_bf16(hi.z), epi_bf16(hi.w)};
          if (c16_vec && col + 7 < g.N) {
            *reinterpret_cast<uint4*>(dst) = make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                                                         e[6] | ((unsigned)e[7] << 16));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (col + j < g.N) dst[j] = e[j];
          }
        }
      }
    }
  }
  if (AMAX && g.amax_out) epi_amax(g, amx);
}
template <int MT, int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue(const addhip_gemm_t& g, f32x16 (&acc)[MT][NT], char* ebuf, int lane, int row0, int col0, int zslab) {
  gemm_epilogue<MT, NT, EPI, false, false>(g, acc, ebuf, lane, row0, col0, zslab, g.alpha);
}

}  // namespace addhip_epi

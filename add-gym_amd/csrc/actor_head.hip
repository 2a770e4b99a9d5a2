// The actor's head section of one optimiser step as ONE kernel (addhip_actor_head): action-mean head forward, PPO clipped-surrogate /
// action-bound / mean-regulariser loss with its gradient, and the backward step through the head into the last hidden layer.
//
// Replaces, per 32-row block of the minibatch and without leaving the chip in between:
//   mean   = H Wh^T + bh                                   DistributionGaussianDiagBuilder.forward (distribution_gaussian_diag.py:47-58)
//   d_mean = d loss / d mean                               PPOAgent._compute_actor_loss (ppo_agent.py:221-275), _compute_action_bound_loss
//                                                          (base_agent.py:522-546), param_reg (distribution_gaussian_diag.py:113-116)
//   dz     = (d_mean Wh) * relu'(H)                        autograd through the head and the last ReLU
//   dWh    = d_mean^T H,  dbh = sum_rows d_mean,  db_top = sum_rows dz
// which the step used to run as three 32-wide GEMM launches (17-25 us each: latency-bound, 20-30 TFLOP/s) + addhip_actor_loss +
// addhip_col_sum.  All three products are v_mfma_f32_32x32x2_f32 on the same 32 x K block of H held in LDS:
//   forward   contraction over k, transposed (mean^T = Wh H^T): A = Wh rows (lane = action dim), B = H rows (lane = row); the four waves take a
//             quarter of K each and add their partial 32x32 tiles through LDS.  The result has lane = row, register = action dim, so the
//             loss's row sums (log-probability, bound and regulariser terms) are register sums + one cross-half shuffle;
//   dz        contraction over the 32 action dims: A = d_mean exactly as those registers hold it (lane = row), B = Wh by columns; the ReLU
//             mask is the H column block already in registers for dWh;
//   dWh       contraction over rows: A = d_mean^T (through a 32x33 LDS tile), B = H read by columns (lane = column); every wave keeps the
//             partial of its K-quarter in registers across all the row blocks of its workgroup and writes ONE slab at the end (combined
//             by addhip_slab_reduce: fixed order).
#include <mutex>
#include "common.h"
#include "record.h"
#include "planes.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int x, int lh) { return (x & 3) + 8 * (x >> 2) + 4 * lh; }  // row of accumulator register x (v_mfma_f32_32x32x*)
__device__ __forceinline__ float half_sum(float v) {  // sum over the 32 lanes of a wave half (the 32 columns of one accumulator row)
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ unsigned short ah_bf16(float v) {
  const unsigned u = __float_as_uint(v);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

constexpr int ah_lds_floats(int K) { return 2 * 32 * (K + 4) + 4 * 1024 + 2 * 32 * 33; }

template <int K>  // width of the last hidden layer: four waves x K/4 columns, K/4 a multiple of 32
__global__ __launch_bounds__(256, 1) void actor_head_kernel(addhip_actor_head_t p) {
  constexpr int LD = K + 4, KS = K / 4, NT = KS / 32, HALF = KS / 2;
  static_assert(KS % 32 == 0, "a wave's share of K is whole 32-column tiles");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;            // [32][LD]  the row block of H
  float* Ws = Hs + 32 * LD;   // [32][LD]  the head weights (rows 29..31 are zero)
  float* red = Ws + 32 * LD;  // [4][16][64] the waves' partial mean tiles, accumulator layout
  float* dm = red + 4 * 1024; // [32][33]  d_mean by rows
  float* dl = dm + 32 * 33;   // [32][33]  d loss / d logstd terms by rows (trainable log-std only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int k0 = wave * KS;
  const int nblk = (p.rows + 31) / 32;
  for (int i = tid; i < 32 * (K / 4); i += 256) {
    const int row = i / (K / 4), c4 = (i % (K / 4)) * 4;
    *reinterpret_cast<float4*>(&Ws[row * LD + c4]) = *reinterpret_cast<const float4*>(&p.Wh[(size_t)row * K + c4]);
  }
  float bias[16], sd[16];  // head bias and standard deviation of action dim acc_row(x, lh)
#pragma unroll
  for (int x = 0; x < 16; ++x) {
    bias[x] = p.bh[acc_row(x, lh)];
    sd[x] = p.dist ? p.dist[acc_row(x, lh)] : p.action_std;
  }
  const float logp_const = p.dist ? p.dist[32] : p.logp_const;
  const float nv = fmaxf(p.n_valid[0], 1.f);
  f32x16 dW[NT];
  float gb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    gb[t] = 0.f;
#pragma unroll
    for (int x = 0; x < 16; ++x) dW[t][x] = 0.f;
  }
  float gbh = 0.f, gls = 0.f, amx = 0.f;
  float st_min = 0.f, st_clip = 0.f, st_ratio = 0.f, st_bound = 0.f, st_reg = 0.f;

  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int r0 = blk * 32;
    __syncthreads();  // the previous block's readers of Hs / dm / red are done (first pass: nothing to wait for)
    for (int i = tid; i < 32 * (K / 4); i += 256) {
      const int row = i / (K / 4), c4 = (i % (K / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + row < p.rows) v = *reinterpret_cast<const float4*>(&p.H[(size_t)(r0 + row) * K + c4]);
      *reinterpret_cast<float4*>(&Hs[row * LD + c4]) = v;
    }
    __syncthreads();
    // ---- forward, TRANSPOSED (mean^T = Wh H^T): this wave's quarter of K; lane half lh walks k0 + HALF*lh .. (any k order: A and B use the
    // same).  The result tile then has lane = ROW and register x = action dim acc_row(x, lh): a row's 32 dims sit in the 16 registers of the
    // two lanes (li, 0) and (li, 1), so the loss's row sums are register sums + ONE cross-half shuffle (the untransposed tile needed 15 per row)
    f32x16 m;
#pragma unroll
    for (int x = 0; x < 16; ++x) m[x] = 0.f;
#pragma unroll 4
    for (int q = 0; q < HALF / 4; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(&Ws[li * LD + k0 + HALF * lh + 4 * q]);
      const float4 b = *reinterpret_cast<const float4*>(&Hs[li * LD + k0 + HALF * lh + 4 * q]);
      m = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, m, 0, 0, 0);
      m = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, m, 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < 16; ++x) red[wave * 1024 + x * 64 + lane] = m[x];
    __syncthreads();
    // ---- loss and d_mean for row r0 + li (every wave, redundantly: each needs d_mean)
    const int row = r0 + li;
    const bool in = row < p.rows;
    const int rr = in ? row : p.rows - 1;
    const bool valid = in && p.rand_mask[rr] == 1.0f;  // ppo_agent.py:229-233
    float na[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // action dims 8c + 4lh .. +3 = acc_row(4c .. 4c+3, lh)
      const float4 v = *reinterpret_cast<const float4*>(&p.norm_action[(size_t)rr * 32 + 8 * c + 4 * lh]);
      na[4 * c] = v.x; na[4 * c + 1] = v.y; na[4 * c + 2] = v.z; na[4 * c + 3] = v.w;
    }
    float mu[16], dd[16], vi[16];
    float sq = 0.f, vs = 0.f, ms = 0.f;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      const int j = acc_row(x, lh);
      const bool col_ok = j < ADDHIP_NUM_DOF;
      const float mu_all = ((red[x * 64 + lane] + red[1024 + x * 64 + lane]) + red[2048 + x * 64 + lane]) + red[3072 + x * 64 + lane] + bias[x];
      mu[x] = col_ok ? mu_all : 0.f;
      dd[x] = col_ok ? (na[x] - mu[x]) / sd[x] : 0.f;
      vi[x] = col_ok ? fminf(mu[x] + 1.f, 0.f) + fmaxf(mu[x] - 1.f, 0.f) : 0.f;  // base_agent.py:536-541 (one of the two is 0)
      sq += dd[x] * dd[x];
      vs += vi[x] * vi[x];
      ms += mu[x] * mu[x];  // param_reg (distribution_gaussian_diag.py:113-116)
    }
    sq += __shfl_xor(sq, 32, 64);
    vs += __shfl_xor(vs, 32, 64);
    ms += __shfl_xor(ms, 32, 64);
    const float logp = -0.5f * sq + logp_const;
    const float ratio = expf(logp - p.old_logp[rr]);
    const float adv = p.adv[rr];
    const float l0 = adv * ratio;
    const float rc = fminf(fmaxf(ratio, 1.f - p.clip_ratio), 1.f + p.clip_ratio);
    const float l1 = adv * rc;
    const bool inrange = ratio >= 1.f - p.clip_ratio && ratio <= 1.f + p.clip_ratio;
    const float gsel = l0 < l1 ? 1.f : (l0 == l1 ? (inrange ? 1.f : 0.5f) : 0.f);  // torch.minimum / clamp subgradients
    const float g_logp = valid ? -(adv * gsel * ratio) / nv : 0.f;
    f32x16 dmean;
#pragma unroll
    for (int x = 0; x < 16; ++x) {
      float g = 0.f;
      if (valid && acc_row(x, lh) < ADDHIP_NUM_DOF) g = g_logp * (dd[x] / sd[x]) + (p.bound_weight * 2.f * vi[x] + p.reg_weight * 2.f * mu[x]) / nv;
      dmean[x] = p.loss_scale * g;
    }
    if (wave == 0) {
      if (valid && lh == 0) {
        st_min += fminf(l0, l1);
        st_clip += fabsf(ratio - 1.f) > p.clip_ratio ? 1.f : 0.f;
        st_ratio += ratio;
        st_bound += vs;
        st_reg += ms;
      }
#pragma unroll
      for (int x = 0; x < 16; ++x) dm[acc_row(x, lh) * 33 + li] = dmean[x];  // d_mean^T [action dim][row]
      if (p.dist) {  // d logp / d logstd_j = dd_j^2 - 1  (distribution_gaussian_diag.py:90-94 differentiated; 0 on the padding dims)
#pragma unroll
        for (int x = 0; x < 16; ++x) dl[acc_row(x, lh) * 33 + li] = acc_row(x, lh) < ADDHIP_NUM_DOF ? p.loss_scale * g_logp * (dd[x] * dd[x] - 1.f) : 0.f;
      }
    }
    __syncthreads();
    if (wave == 0 && lh == 0) {  // d bh[li] += sum over the block's rows
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 32; ++r) s += dm[li * 33 + r];
      gbh += s;
      if (p.dist) {
        float sl = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) sl += dl[li * 33 + r];
        gls += sl;
      }
    }
    float dT[16];  // d_mean^T fragments of the weight-gradient product: action dim li, rows acc_row(x, lh)
#pragma unroll
    for (int x = 0; x < 16; ++x) dT[x] = dm[li * 33 + acc_row(x, lh)];
    // ---- this wave's K-quarter, 32 columns at a time: dWh += d_mean^T H ; dz = (d_mean Wh) * relu'(H)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int kc = k0 + 32 * t + li;
      float h[16];
#pragma unroll
      for (int x = 0; x < 16; ++x) h[x] = Hs[acc_row(x, lh) * LD + kc];
#pragma unroll
      for (int x = 0; x < 16; ++x) dW[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dT[x], h[x], dW[t], 0, 0, 0);
      f32x16 z;
#pragma unroll
      for (int x = 0; x < 16; ++x) z[x] = 0.f;
      // (contraction over the action dims in the order the registers hold them: step x takes dims acc_row(x, 0) and acc_row(x, 1))
#pragma unroll
      for (int x = 0; x < 16; ++x) z = __builtin_amdgcn_mfma_f32_32x32x2f32(dmean[x], Ws[acc_row(x, lh) * LD + kc], z, 0, 0, 0);
      float cs = 0.f;
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const float v = h[x] > 0.f ? z[x] : 0.f;
        const int orow = r0 + acc_row(x, lh);
        cs += v;
        amx = fmaxf(amx, fabsf(v));
        if (orow < p.rows) {
          if (p.dz) p.dz[(size_t)orow * K + kc] = v;
          if (p.dz16) {
            if (p.planes16 == ADDHIP_STORE_BF16X3) addhip_planes::store1(p.dz16 + 3 * (size_t)orow * K, kc, v);
            else p.dz16[(size_t)orow * K + kc] = ah_bf16(v);
          }
        }
      }
      gb[t] += cs + __shfl_xor(cs, 32, 64);
    }
  }
  // ---- this workgroup's partial of (dWh | dbh) -> its slab; db_top -> its replica row; loss diagnostics
  float* slab = p.slabs + (size_t)blockIdx.x * ADDHIP_ACTOR_HEAD_SLAB(K);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int kc = k0 + 32 * t + li;
#pragma unroll
    for (int x = 0; x < 16; ++x) slab[acc_row(x, lh) * K + kc] = dW[t][x];
    if (lh == 0 && p.gb_top) atomicAdd(&p.gb_top[(size_t)(blockIdx.x % p.gb_replicas) * p.ld_gb + kc], gb[t]);
  }
  if (wave == 0 && lh == 0) {  // (lane li: action dim li)
    slab[32 * K + li] = gbh;
    slab[32 * K + 32 + li] = gls;
  }
  if (p.amax) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
    if (lane == 0) atomicMax(&p.amax[blockIdx.x % ADDHIP_AMAX_SLOTS], __float_as_uint(amx));
  }
  if (wave == 0) {  // (lanes 0..31 hold the sums over their rows)
    st_min = half_sum(st_min); st_clip = half_sum(st_clip); st_ratio = half_sum(st_ratio); st_bound = half_sum(st_bound); st_reg = half_sum(st_reg);
    if (lane == 0) {  // per-minibatch MEANS over the exploring samples (ppo_agent.py:229-247)
      atomicAdd(&p.stats[0], st_min / nv); atomicAdd(&p.stats[1], st_clip / nv); atomicAdd(&p.stats[2], st_ratio / nv); atomicAdd(&p.stats[3], st_bound / nv);
      if (p.reg_weight != 0.f) atomicAdd(&p.stats[5], st_reg / nv);
    }
  }
}

template <int K>
int launch_head(const addhip_actor_head_t& p, int grid, hipStream_t st) {
  static std::mutex mu;
  static bool done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (!done[dev]) {
      ADDHIP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(actor_head_kernel<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * ah_lds_floats(K))));
      done[dev] = true;
    }
  }
  hipLaunchKernelGGL(actor_head_kernel<K>, dim3(grid), dim3(256), sizeof(float) * ah_lds_floats(K), st, p);
  return addhip::check_launch("actor_head_kernel");
}

}  // namespace

extern "C" int addhip_actor_head_slabs(int32_t rows) {
  const int nblk = (rows + 31) / 32;
  return nblk < 256 ? (nblk < 1 ? 1 : nblk) : 256;
}

extern "C" int addhip_actor_head(const addhip_actor_head_t* pp, void* stream) {
  ADDHIP_REQUIRE(pp, "actor_head: null descriptor");
  const addhip_actor_head_t p = *pp;
  ADDHIP_REQUIRE(p.rows > 0 && (p.hidden == 128 || p.hidden == 256 || p.hidden == 512), "actor_head: hidden width 128, 256 or 512 (got %d)", p.hidden);
  ADDHIP_REQUIRE(p.H && p.Wh && p.bh && p.norm_action && p.old_logp && p.adv && p.rand_mask && p.n_valid && p.stats, "actor_head: input pointers missing");
  ADDHIP_REQUIRE((p.dz || p.dz16) && p.slabs && p.num_slabs == addhip_actor_head_slabs(p.rows), "actor_head: outputs missing, or num_slabs != addhip_actor_head_slabs(rows)");
  ADDHIP_REQUIRE(aligned16(p.H) && aligned16(p.Wh) && (!p.dz16 || p.planes16 != ADDHIP_STORE_BF16X3 || aligned16(p.dz16)), "actor_head: misaligned buffers");
  ADDHIP_REQUIRE(!p.gb_top || (p.gb_replicas >= 1 && p.ld_gb >= p.hidden), "actor_head: gb_top needs gb_replicas >= 1 rows of ld_gb >= hidden floats");
  ADDHIP_RECORDABLE(addhip_actor_head, pp);
  hipStream_t st = (hipStream_t)stream;
  return p.hidden == 512 ? launch_head<512>(p, p.num_slabs, st) : p.hidden == 256 ? launch_head<256>(p, p.num_slabs, st) : launch_head<128>(p, p.num_slabs, st);
}

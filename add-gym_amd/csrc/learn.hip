// Learner-side kernels of the hot path that are not GEMMs: actor sampling, Philox fills, discriminator
// reward / sampler statistics, TD(lambda) + advantage normalisation, running normalisers, minibatch gather,
// loss heads (PPO clip, critic MSE, ADD discriminator BCE + gradient penalty), AdamW.  All HBM-bound:
// one wavefront per row with 64-lane shuffle reductions, or one thread per element, coalesced.
// Reference functions restated: see include/addhip.h at each entry point.
#include "common.h"
#include "planes.h"
#include "record.h"
#include "philox.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over a 256-thread block (result valid on thread 0)
__device__ __forceinline__ float block_sum(float v, float* sh /*[4]*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// fp32 -> bf16, round to nearest even (finite inputs), as addhip_to_bf16
__device__ __forceinline__ unsigned short bf16_rne(float v) {
  const unsigned u = __float_as_uint(v);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ uint2 bf16_pack4(float4 o) {
  return make_uint2((unsigned)bf16_rne(o.x) | ((unsigned)bf16_rne(o.y) << 16), (unsigned)bf16_rne(o.z) | ((unsigned)bf16_rne(o.w) << 16));
}

// running max |v| of a thread -> one atomic per wave on slot (block % ADDHIP_AMAX_SLOTS) (the operand bounds of ADDHIP_PREC_F16X2 GEMMs)
__device__ __forceinline__ void amax_commit(unsigned* slots, float amx) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(&slots[blockIdx.x % ADDHIP_AMAX_SLOTS], __float_as_uint(amx));
}
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long count, unsigned* slots) {
  float amx = 0.f;
  const long long n4 = count >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) amx = fmaxf(amx, fabsf(x[(n4 << 2) + threadIdx.x]));
  amax_commit(slots, amx);
}

inline int row_grid(long long rows) {  // 4 waves (rows) per 256-thread block, grid-stride
  long long g = (rows + 3) / 4;
  return (int)(g < 2048 ? g : 2048);
}
inline int elem_grid(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g < 4096 ? g : 4096);
}

// (sid_base: optional device counter added to the stream id -- lets a captured hipGraph draw fresh numbers on every replay)
__global__ void fill_uniform_kernel(float* out, long long n, uint64_t seed, uint64_t sid, const uint64_t* sid_base) {
  if (sid_base) sid += sid_base[0];
  long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
  for (; q * 4 < n; q += step) {
    U4 r = philox((uint64_t)q, sid, seed);
    float v[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < n) out[q * 4 + k] = v[k];
  }
}
__global__ void fill_normal_kernel(float* out, long long n, uint64_t seed, uint64_t sid, const uint64_t* sid_base) {
  if (sid_base) sid += sid_base[0];
  long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
  for (; q * 4 < n; q += step) {
    U4 r = philox((uint64_t)q, sid, seed);
    float u1 = 1.0f - u01(r.x), u2 = u01(r.y), u3 = 1.0f - u01(r.z), u4 = u01(r.w);  // (0,1]
    float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
    float v[4] = {ra * cosf(6.2831853071795865f * u2), ra * sinf(6.2831853071795865f * u2),
                  rb * cosf(6.2831853071795865f * u4), rb * sinf(6.2831853071795865f * u4)};
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < n) out[q * 4 + k] = v[k];
  }
}

// ------------------------------------------------------------------ rollout actor head
// log-std vector -> {std, log-probability constant, entropy} (distribution_gaussian_diag.py:63-67, 90-99): one wave
__global__ __launch_bounds__(64) void dist_refresh_kernel(const float* logstd, float* dist) {
  const int lane = threadIdx.x;
  const float ls = lane < ADDHIP_NUM_DOF ? logstd[lane] : 0.f;
  if (lane < 32) dist[lane] = lane < ADDHIP_NUM_DOF ? expf(ls) : 1.f;
  const float s = wave_sum(ls);
  if (lane == 0) {
    dist[32] = __fsub_rn((float)(-0.5 * ADDHIP_NUM_DOF * 1.8378770664093453), s);        // -0.5 * 29 * log(2 pi) - sum(logstd)
    dist[33] = __fadd_rn(s, (float)(0.5 * ADDHIP_NUM_DOF * 2.8378770664093453));         // sum(logstd) + 0.5 * 29 * log(2 pi e)
  }
}

__global__ __launch_bounds__(256) void actor_sample_kernel(const float* mean, int ldm, const float* noise, float stdv, float logp_const, const float* dist,
                                                           const float* logstd_rows, const float* a_mean, const float* a_std, int n, int deterministic_all,
                                                           const float* explore_u, float exp_prob, float* action, float* a_logp, float* rand_mask) {
  const int lane = threadIdx.x & 63;
  if (dist) {  // trainable log-std: a standard deviation per action dimension
    stdv = dist[lane & 31];
    logp_const = dist[32];
  }
  for (int env = blockIdx.x * 4 + (threadIdx.x >> 6); env < n; env += gridDim.x * 4) {
    // rand_action_mask = bernoulli(exp_prob) per env (ppo_agent.py:80-88): a uniform draw below the probability explores
    const bool deterministic = deterministic_all || (explore_u && !(explore_u[env] < exp_prob));
    float sq = 0.f, act = 0.f;
    if (logstd_rows) {  // actor_std_type VARIABLE: this sample's own log-std row (distribution_gaussian_diag.py:52-53, 63-67, 90-94)
      const float ls = lane < ADDHIP_NUM_DOF ? logstd_rows[(size_t)env * ldm + lane] : 0.f;
      stdv = expf(ls);
      logp_const = __fsub_rn((float)(-0.5 * ADDHIP_NUM_DOF * 1.8378770664093453), wave_sum(ls));
    }
    if (lane < ADDHIP_NUM_DOF) {
      float mu = mean[(size_t)env * ldm + lane];
      float na = deterministic ? mu : __fadd_rn(mu, __fmul_rn(stdv, noise[(size_t)env * ADDHIP_NUM_DOF + lane]));
      float d = __fdiv_rn(__fsub_rn(na, mu), stdv);
      sq = __fmul_rn(d, d);
      act = __fadd_rn(__fmul_rn(na, a_std[lane]), a_mean[lane]);  // Normalizer.unnormalize
    }
    sq = wave_sum(sq);
    if (lane < 32) action[(size_t)env * 32 + lane] = act;
    if (lane == 0) {
      a_logp[env] = __fadd_rn(__fmul_rn(-0.5f, sq), logp_const);
      rand_mask[env] = deterministic ? 0.f : 1.f;
    }
  }
}

// ------------------------------------------------------------------ build-train-data
__global__ __launch_bounds__(256) void disc_prep_kernel(const float* dobs, const float* ddemo, int stride, int dim, long long rows,
                                                        const float* mean_abs, float min_diff, float* norm_diff, const int* motion_id,
                                                        const float* motion_time, addhip_sampler_t s, int cells, float* abs_sum) {
  extern __shared__ float sh[];  // [cells] err sums | [cells] counts | [4][stride] abs partials
  float* sh_sum = sh;
  float* sh_cnt = sh + cells;
  float* sh_abs = sh + 2 * cells;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * cells; i += 256) sh[i] = 0.f;
  float a[4] = {0.f, 0.f, 0.f, 0.f};  // |diff| partials for columns lane + 64k
  __syncthreads();
  for (long long row = (long long)blockIdx.x * 4 + w; row < rows; row += (long long)gridDim.x * 4) {
    float err = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane + 64 * k;
      if (c >= stride) break;
      float v = 0.f;
      if (c < dim) {
        float ag = dobs[row * stride + c], de = ddemo[row * stride + c];
        float d = de - ag;
        float e = ag - de;
        err += e * e;
        v = d / fmaxf(mean_abs[c], min_diff);
        a[k] += fabsf(d);
      }
      if (norm_diff) norm_diff[row * stride + c] = v;
    }
    err = wave_sum(err);
    if (lane == 0 && motion_id) {  // AdaptiveSegmentSampler.update_errors index math (sampler.py:27-35)
      int id = motion_id[row];
      float seg = fmaxf(s.seg_size[id], 1e-6f);
      long long si = (long long)(motion_time[row] / seg);
      si = si < 0 ? 0 : (si > s.num_segments - 1 ? s.num_segments - 1 : si);
      atomicAdd(&sh_sum[id * s.num_segments + (int)si], err);
      atomicAdd(&sh_cnt[id * s.num_segments + (int)si], 1.0f);
    }
  }
  if (abs_sum) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (lane + 64 * k < stride) sh_abs[w * stride + lane + 64 * k] = a[k];
  }
  __syncthreads();
  if (abs_sum)
    for (int c = threadIdx.x; c < dim; c += 256)
      atomicAdd(&abs_sum[c], sh_abs[c] + sh_abs[stride + c] + sh_abs[2 * stride + c] + sh_abs[3 * stride + c]);
  if (motion_id)
    for (int i = threadIdx.x; i < cells; i += 256)
      if (sh_cnt[i] > 0.f) {
        atomicAdd(&s.err_sum[i], sh_sum[i]);
        atomicAdd(&s.err_cnt[i], sh_cnt[i]);
      }
}

__global__ void sampler_update_kernel(addhip_sampler_t s, int cells) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells) return;
  float c = s.err_cnt[i];
  if (c > 0.f) s.errors[i] = 0.9f * s.errors[i] + 0.1f * (s.err_sum[i] / c);  // sampler.py:52-55
  s.err_sum[i] = 0.f;
  s.err_cnt[i] = 0.f;
}

__global__ __launch_bounds__(256) void disc_reward_kernel(const float* logits, float* reward, long long n, float scale, float task_w,
                                                          float disc_w, float* stats) {
  __shared__ float sh[4];
  float s1 = 0.f, s2 = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float prob = 1.0f / (1.0f + expf(-logits[i]));
    float dr = -logf(fmaxf(1.0f - prob, 0.0001f)) * scale;  // amp_agent.py:200-205
    reward[i] = task_w * reward[i] + disc_w * dr;           // add_agent.py:124
    s1 += dr;
    s2 += dr * dr;
  }
  float t1 = block_sum(s1, sh);
  float t2 = block_sum(s2, sh);
  if (threadIdx.x == 0 && stats) {
    atomicAdd(&stats[0], t1);
    atomicAdd(&stats[1], t2);
  }
}

__global__ __launch_bounds__(256) void head_gemv_kernel(const float* H, int ld, int K, long long rows, const float* w, const float* b, float* out) {
  const int lane = threadIdx.x & 63;
  for (long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (long long)gridDim.x * 4) {
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      float4 h = *reinterpret_cast<const float4*>(H + row * ld + k);
      float4 ww = *reinterpret_cast<const float4*>(w + k);
      acc += h.x * ww.x + h.y * ww.y + h.z * ww.z + h.w * ww.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc + b[0];
  }
}

// ------------------------------------------------------------------ TD(lambda) + advantage
__global__ __launch_bounds__(256) void td_lambda_kernel(const float* reward, const float* next_vals, const float* timeout_vals, const float* vals,
                                                        const int* done, const float* rand_mask, int T, int N, float discount, float lam,
                                                        float succ_val, float fail_val, float* tar_val, float* adv, double* partial) {
  __shared__ double shd[3][4];
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  double s1 = 0.0, s2 = 0.0, cnt = 0.0;
  if (n < N) {
    float next_ret = 0.f;
    for (int t = T - 1; t >= 0; --t) {
      const size_t i = (size_t)t * N + n;
      const int d = done[i];
      float nv = next_vals[i];
      if (d == ADDHIP_DONE_TIME && timeout_vals) nv = timeout_vals[n];  // V(true next obs); next_vals row holds the reset obs
      if (d == ADDHIP_DONE_SUCC) nv = succ_val;  // ppo_agent.py:127-133
      if (d == ADDHIP_DONE_FAIL) nv = fail_val;
      float ret;
      if (t == T - 1) {
        ret = __fadd_rn(reward[i], __fmul_rn(discount, nv));  // base_agent.py:632-633
      } else {
        float reset = d != ADDHIP_DONE_NULL ? 1.f : 0.f;
        float cl = __fmul_rn(lam, __fsub_rn(1.f, reset));
        float mix = __fadd_rn(__fmul_rn(__fsub_rn(1.f, cl), nv), __fmul_rn(cl, next_ret));
        ret = __fadd_rn(reward[i], __fmul_rn(discount, mix));  // base_agent.py:641-644
      }
      next_ret = ret;
      tar_val[i] = ret;
      float a = __fsub_rn(ret, vals[i]);
      adv[i] = a;
      if (rand_mask[i] == 1.0f) { s1 += a; s2 += (double)a * a; cnt += 1.0; }
    }
  }
  s1 = wave_sum_d(s1); s2 = wave_sum_d(s2); cnt = wave_sum_d(cnt);
  if ((threadIdx.x & 63) == 0) { shd[0][threadIdx.x >> 6] = s1; shd[1][threadIdx.x >> 6] = s2; shd[2][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[3 * blockIdx.x + 0] = shd[0][0] + shd[0][1] + shd[0][2] + shd[0][3];
    partial[3 * blockIdx.x + 1] = shd[1][0] + shd[1][1] + shd[1][2] + shd[1][3];
    partial[3 * blockIdx.x + 2] = shd[2][0] + shd[2][1] + shd[2][2] + shd[2][3];
  }
}
__global__ void adv_stats_kernel(const double* partial, int blocks, float* stats_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s1 = 0, s2 = 0, c = 0;
  for (int b = 0; b < blocks; ++b) { s1 += partial[3 * b]; s2 += partial[3 * b + 1]; c += partial[3 * b + 2]; }
  double mean = c > 0 ? s1 / c : 0.0;
  double var = c > 1 ? (s2 - c * mean * mean) / (c - 1.0) : 0.0;  // unbiased (torch.std_mean, ppo_agent.py:147)
  stats_out[0] = (float)mean;
  stats_out[1] = (float)sqrt(var > 0 ? var : 0.0);
}
__global__ void adv_norm_kernel(float* adv, long long n, const float* stats, float clip) {
  const float mean = stats[0], sd = fmaxf(stats[1], 1e-5f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = __fdiv_rn(__fsub_rn(adv[i], mean), sd);
    adv[i] = fminf(fmaxf(v, -clip), clip);  // ppo_agent.py:149-150
  }
}

// ------------------------------------------------------------------ normalisers
__global__ __launch_bounds__(256) void norm_accum_kernel(const float* X, long long rows, int dim, int ld, float* sum, float* sumsq, long long rows_per_block) {
  __shared__ float p1[4][64], p2[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s1 = 0.f, s2 = 0.f;
  if (c < dim)
    for (long long r = r0 + w; r < r1; r += 4) { float x = X[r * ld + c]; s1 += x; s2 += x * x; }
  p1[w][threadIdx.x & 63] = s1; p2[w][threadIdx.x & 63] = s2;
  __syncthreads();
  if (w == 0 && c < dim) {
    atomicAdd(&sum[c], p1[0][threadIdx.x] + p1[1][threadIdx.x] + p1[2][threadIdx.x] + p1[3][threadIdx.x]);
    if (sumsq) atomicAdd(&sumsq[c], p2[0][threadIdx.x] + p2[1][threadIdx.x] + p2[2][threadIdx.x] + p2[3][threadIdx.x]);
  }
}
__global__ void norm_merge_kernel(float* mean, float* stdv, float* mean_sq, long long* count, float* sum, float* sumsq, long long new_count,
                                  int dim, float min_var, int first) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const long long old = *count;
  __syncthreads();
  if (i < dim && new_count > 0) {
    float msq = first ? stdv[i] * stdv[i] + mean[i] * mean[i] : mean_sq[i];  // normalizer.py:38-39,134-137
    const long long total = old + new_count;
    const float w_old = (float)old / (float)total, w_new = (float)new_count / (float)total;
    float nm = sum[i] / (float)new_count, nsq = sumsq[i] / (float)new_count;
    float m = w_old * mean[i] + w_new * nm;
    msq = w_old * msq + w_new * nsq;
    mean[i] = m;
    mean_sq[i] = msq;
    stdv[i] = sqrtf(fmaxf(msq - m * m, min_var));
    sum[i] = 0.f;
    sumsq[i] = 0.f;
  }
  __syncthreads();
  if (i == 0 && new_count > 0) *count = old + new_count;
}
__global__ void diffnorm_merge_kernel(float* mean_abs, long long* count, float* abs_sum, long long new_count, int dim) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const long long old = *count;
  __syncthreads();
  if (i < dim) {
    const long long total = old + new_count;
    const float w_old = (float)old / (float)total, w_new = (float)new_count / (float)total;
    mean_abs[i] = w_old * mean_abs[i] + w_new * (abs_sum[i] / (float)new_count);  // diff_normalizer.py:33-45
    abs_sum[i] = 0.f;
  }
  __syncthreads();
  if (i == 0) *count = old + new_count;
}

// ------------------------------------------------------------------ minibatch gather
__global__ __launch_bounds__(256) void gather_kernel(addhip_gather_t g) {
  const int lane = threadIdx.x & 63;
  float amx_o = 0.f, amx_d = 0.f;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < g.count; r += gridDim.x * 4) {
    const long long src = g.idx[r];
    for (int c = lane; c < g.obs_stride; c += 64) {
      const float v = c < g.obs_dim ? (g.obs[src * g.obs_stride + c] - g.obs_mean[c]) / g.obs_std[c] : 0.f;
      g.norm_obs[(size_t)r * g.obs_stride + c] = v;
      amx_o = fmaxf(amx_o, fabsf(v));
      if (g.norm_obs16) {
        if (g.planes16 == ADDHIP_STORE_BF16X3) addhip_planes::store1(g.norm_obs16 + 3 * (size_t)r * g.obs_stride, c, v);
        else g.norm_obs16[(size_t)r * g.obs_stride + c] = bf16_rne(v);
      }
    }
    if (lane < 32)
      g.norm_action[(size_t)r * 32 + lane] = lane < ADDHIP_NUM_DOF ? (g.action[src * 32 + lane] - g.a_mean[lane]) / g.a_std[lane] : 0.f;
    for (int c = lane; c < g.disc_stride; c += 64) {
      float v = 0.f;
      if (c < g.disc_dim) v = (g.disc_demo[src * g.disc_stride + c] - g.disc_obs[src * g.disc_stride + c]) / fmaxf(g.mean_abs[c], g.min_diff);
      g.norm_diff[(size_t)r * g.disc_stride + c] = v;
      amx_d = fmaxf(amx_d, fabsf(v));
      if (g.norm_diff16) {
        if (g.planes16 == ADDHIP_STORE_BF16X3) addhip_planes::store1(g.norm_diff16 + 3 * (size_t)r * g.disc_stride, c, v);
        else g.norm_diff16[(size_t)r * g.disc_stride + c] = bf16_rne(v);
      }
    }
    if (lane == 0) {
      g.o_logp[r] = g.a_logp[src];
      g.o_adv[r] = g.adv[src];
      g.o_tar_val[r] = g.tar_val[src];
      g.o_mask[r] = g.rand_mask[src];
    }
  }
  if (g.obs_amax) amax_commit(g.obs_amax, amx_o);
  if (g.diff_amax) amax_commit(g.diff_amax, amx_d);
}

// ------------------------------------------------------------------ loss heads
__global__ __launch_bounds__(256) void count_mask_kernel(const float* mask, int M, float* out) {
  __shared__ float sh[4];
  float c = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) c += mask[i] == 1.0f ? 1.f : 0.f;
  float t = block_sum(c, sh);
  if (threadIdx.x == 0) atomicAdd(out, t);
}

__global__ __launch_bounds__(256) void actor_loss_kernel(const float* mean, const float* na, const float* old_logp, const float* adv,
                                                         const float* mask, int M, float stdv, float logp_const, const float* dist, float clip, float bound_w,
                                                         float reg_w, float loss_scale, const float* n_valid, float* d_mean, float* g_logstd, float* stats, int ldm,
                                                         const float* logstd_rows, float ent_w) {
  __shared__ float sh[4];
  __shared__ float sh_ls[4][32];
  const int lane = threadIdx.x & 63;
  const float nv = fmaxf(n_valid[0], 1.f);
  if (dist) {
    stdv = dist[lane & 31];
    logp_const = dist[32];
  }
  float gls = 0.f;  // d loss / d logstd[lane], this wave's rows
  float st_min = 0.f, st_clip = 0.f, st_ratio = 0.f, st_bound = 0.f, st_reg = 0.f, st_ent = 0.f;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < M; r += gridDim.x * 4) {
    const bool valid = mask[r] == 1.0f;  // ppo_agent.py:229-233
    float mu = 0.f, d = 0.f, viol = 0.f;
    if (logstd_rows) {  // actor_std_type VARIABLE: per-sample log-std (the columns behind the mean's)
      const float ls = lane < ADDHIP_NUM_DOF ? logstd_rows[(size_t)r * ldm + lane] : 0.f;
      stdv = expf(ls);
      const float ls_sum = wave_sum(ls);
      logp_const = (float)(-0.5 * ADDHIP_NUM_DOF * 1.8378770664093453) - ls_sum;
      if (valid && lane == 0) st_ent += ls_sum + (float)(0.5 * ADDHIP_NUM_DOF * 2.8378770664093453);  // entropy of this sample's distribution (:96-99)
    }
    if (lane < ADDHIP_NUM_DOF) {
      mu = mean[(size_t)r * ldm + lane];
      d = (na[(size_t)r * 32 + lane] - mu) / stdv;
      viol = fminf(mu + 1.f, 0.f) + fmaxf(mu - 1.f, 0.f);  // base_agent.py:536-541 (one of the two is 0)
    }
    float sq = wave_sum(d * d);
    float vs = wave_sum(viol * viol);
    float ms = reg_w != 0.f ? wave_sum(mu * mu) : 0.f;  // DistributionGaussianDiag.param_reg (distribution_gaussian_diag.py:113-116)
    float logp = -0.5f * sq + logp_const;
    float ratio = expf(logp - old_logp[r]);
    float a = adv[r];
    float l0 = a * ratio;
    float rc = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
    float l1 = a * rc;
    bool inrange = ratio >= 1.f - clip && ratio <= 1.f + clip;
    float gsel = l0 < l1 ? 1.f : (l0 == l1 ? (inrange ? 1.f : 0.5f) : 0.f);  // torch.minimum / clamp subgradients
    float g_logp = valid ? -(a * gsel * ratio) / nv : 0.f;
    if (lane < 32) {
      float g = 0.f;
      if (valid && lane < ADDHIP_NUM_DOF) g = g_logp * (d / stdv) + (bound_w * 2.f * viol + reg_w * 2.f * mu) / nv;
      d_mean[(size_t)r * ldm + lane] = loss_scale * g;
      // d logp / d logstd_j = d_j^2 - 1;  per-sample log-std: + the entropy bonus -ent_w * mean(entropy), whose gradient is -ent_w / nv on every log-std of an exploring sample
      const float dls = lane < ADDHIP_NUM_DOF ? loss_scale * (g_logp * (d * d - 1.f) - ((logstd_rows && valid) ? ent_w / nv : 0.f)) : 0.f;
      gls += dls;
      if (logstd_rows) d_mean[(size_t)r * ldm + 32 + lane] = dls;  // (VARIABLE: the second head's output gradient, columns 32..63)
    }
    if (valid && lane == 0) {
      st_min += fminf(l0, l1);
      st_clip += fabsf(ratio - 1.f) > clip ? 1.f : 0.f;
      st_ratio += ratio;
      st_bound += vs;
      st_reg += ms;
    }
  }
  float t0 = block_sum(st_min, sh), t1 = block_sum(st_clip, sh), t2 = block_sum(st_ratio, sh), t3 = block_sum(st_bound, sh);
  float t5 = block_sum(st_reg, sh);
  if (logstd_rows && ent_w != 0.f) {
    const float t6 = block_sum(st_ent, sh);
    if (threadIdx.x == 0) atomicAdd(&stats[6], t6 / nv);
  }
  if (dist && g_logstd) {
    if (lane < 32) sh_ls[threadIdx.x >> 6][lane] = gls;
    __syncthreads();
    if (threadIdx.x < ADDHIP_NUM_DOF) atomicAdd(&g_logstd[threadIdx.x], (sh_ls[0][threadIdx.x] + sh_ls[1][threadIdx.x]) + (sh_ls[2][threadIdx.x] + sh_ls[3][threadIdx.x]));
  }
  if (threadIdx.x == 0) {  // per-minibatch MEANS over the exploring samples (ppo_agent.py:229-247): nv varies with exp_prob < 1
    atomicAdd(&stats[0], t0 / nv); atomicAdd(&stats[1], t1 / nv); atomicAdd(&stats[2], t2 / nv); atomicAdd(&stats[3], t3 / nv);
    if (reg_w != 0.f) atomicAdd(&stats[5], t5 / nv);
  }
}

__global__ __launch_bounds__(256) void critic_head_kernel(const float* H, int ld, int K, int M, const float* w, const float* b, const float* tar,
                                                          float loss_scale, float* dZ, float* dv_out, float* stats) {
  __shared__ float sh[4];
  const int lane = threadIdx.x & 63;
  float se = 0.f;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < M; r += gridDim.x * 4) {
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      float4 h = *reinterpret_cast<const float4*>(H + (size_t)r * ld + k);
      float4 ww = *reinterpret_cast<const float4*>(w + k);
      acc += h.x * ww.x + h.y * ww.y + h.z * ww.z + h.w * ww.w;
    }
    float v = wave_sum(acc) + b[0];
    float diff = tar[r] - v;  // ppo_agent.py:215-216
    float dv = loss_scale * 2.f * (v - tar[r]) / (float)M;
    if (dZ)  // (NULL: addhip_head_backward produces it together with the head's gradients)
      for (int k = lane * 4; k < K; k += 256) {
        float4 h = *reinterpret_cast<const float4*>(H + (size_t)r * ld + k);
        float4 ww = *reinterpret_cast<const float4*>(w + k);
        float4 o = make_float4(h.x > 0.f ? dv * ww.x : 0.f, h.y > 0.f ? dv * ww.y : 0.f, h.z > 0.f ? dv * ww.z : 0.f, h.w > 0.f ? dv * ww.w : 0.f);
        *reinterpret_cast<float4*>(dZ + (size_t)r * ld + k) = o;
      }
    if (lane == 0) { dv_out[r] = dv; se += diff * diff; }
  }
  float t = block_sum(se, sh);
  if (threadIdx.x == 0) atomicAdd(&stats[0], t);
}

__device__ __forceinline__ float bce_logits(float x, float y) { return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void disc_head_kernel(const float* H, int ld, int K, int M, const float* h_pos, const float* w, const float* b,
                                                        float loss_scale, float* dlogit, float* dlogit_pos, float* stats) {
  __shared__ float sh[4];
  const int lane = threadIdx.x & 63;
  float s_bce = 0.f, s_logit = 0.f, s_acc = 0.f;
  // row M is the single zero-difference "positive" sample (add_agent.py:145-149)
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r <= M; r += gridDim.x * 4) {
    const float* row = r < M ? H + (size_t)r * ld : h_pos;
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      float4 h = *reinterpret_cast<const float4*>(row + k);
      float4 ww = *reinterpret_cast<const float4*>(w + k);
      acc += h.x * ww.x + h.y * ww.y + h.z * ww.z + h.w * ww.w;
    }
    float x = wave_sum(acc) + b[0];
    if (lane == 0) {
      if (r < M) {
        dlogit[r] = loss_scale * 0.5f * (sigmoidf(x) - 0.1f) / (float)M;  // amp_agent.py:177-180
        s_bce += bce_logits(x, 0.1f); s_logit += x; s_acc += x < 0.f ? 1.f : 0.f;
      } else {
        dlogit_pos[0] = loss_scale * 0.5f * (sigmoidf(x) - 0.9f);          // amp_agent.py:182-185
        atomicAdd(&stats[1], bce_logits(x, 0.9f));
        atomicAdd(&stats[3], x);
        atomicAdd(&stats[5], x > 0.f ? 1.f : 0.f);
      }
    }
  }
  float t0 = block_sum(s_bce, sh), t2 = block_sum(s_logit, sh), t4 = block_sum(s_acc, sh);
  if (threadIdx.x == 0) { atomicAdd(&stats[0], t0); atomicAdd(&stats[2], t2); atomicAdd(&stats[4], t4); }
}

__global__ void outer_mask_kernel(const float* v, const float* w, const float* H, int ld, int K, long long rows, float* out, unsigned short* out16, int planes16, unsigned* amax) {
  const long long n = rows * (K / 4);
  float amx = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    long long r = i / (K / 4);
    int k = (int)(i - r * (K / 4)) * 4;
    float4 h = *reinterpret_cast<const float4*>(H + r * ld + k);
    float4 ww = *reinterpret_cast<const float4*>(w + k);
    float s = v ? v[r] : 1.f;
    float4 o = make_float4(h.x > 0.f ? s * ww.x : 0.f, h.y > 0.f ? s * ww.y : 0.f, h.z > 0.f ? s * ww.z : 0.f, h.w > 0.f ? s * ww.w : 0.f);
    if (out) *reinterpret_cast<float4*>(out + r * ld + k) = o;
    if (out16) {
      if (planes16 == ADDHIP_STORE_BF16X3) addhip_planes::store4(out16 + 3 * r * ld, k, o);
      else *reinterpret_cast<uint2*>(out16 + r * ld + k) = bf16_pack4(o);
    }
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
  }
  if (amax) amax_commit(amax, amx);  // (uniform)
}

// One pass over the last hidden layer's activations H for everything a scalar head needs in the backward direction:
//   dZ[r,k]  = (H[r,k] > 0) ? v[r] * w[k] : 0        gradient w.r.t. the layer's pre-activation (what outer_mask writes)
//   dW[k]   += sum_r v[r] * H[r,k]                    head weight gradient       (weighted_col_sum)
//   db[0]   += sum_r v[r]                             head bias gradient         (col_sum of v)
//   dbt[k]  += sum_r dZ[r,k]                          the layer's bias gradient  (col_sum of dZ)
// A lane owns the same columns for every row its wave visits, so the column sums are per-lane registers; the four waves
// of a workgroup are combined in LDS and each workgroup issues one atomic per column.  K <= 1024.
__global__ __launch_bounds__(256) void head_backward_kernel(const float* v, const float* w, const float* H, int ld, int K, long long rows, float* dZ,
                                                            unsigned short* dZ16, int planes16, float* dW, float* db, float* dbt, unsigned* amax, float* ordered) {
  __shared__ float red[4][1024];
  __shared__ float sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 gw[4], gb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) gw[j] = gb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float sv = 0.f, amx = 0.f;
  for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
    const float s = v[r];
    if (lane == 0) sv += s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = lane * 4 + 256 * j;
      if (k < K) {
        const float4 h = *reinterpret_cast<const float4*>(H + r * ld + k);
        const float4 ww = *reinterpret_cast<const float4*>(w + k);
        const float4 o = make_float4(h.x > 0.f ? s * ww.x : 0.f, h.y > 0.f ? s * ww.y : 0.f, h.z > 0.f ? s * ww.z : 0.f, h.w > 0.f ? s * ww.w : 0.f);
        if (dZ) *reinterpret_cast<float4*>(dZ + r * ld + k) = o;
        if (dZ16) {
          if (planes16 == ADDHIP_STORE_BF16X3) addhip_planes::store4(dZ16 + 3 * r * ld, k, o);
          else *reinterpret_cast<uint2*>(dZ16 + r * ld + k) = bf16_pack4(o);
        }
        gw[j].x += s * h.x; gw[j].y += s * h.y; gw[j].z += s * h.z; gw[j].w += s * h.w;
        gb[j].x += o.x; gb[j].y += o.y; gb[j].z += o.z; gb[j].w += o.w;
        amx = fmaxf(fmaxf(amx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
      }
    }
  }
  // ordered != NULL: this workgroup's partial sums go to ordered[blockIdx.x][0..K) (dW), [K..2K) (dbt), [2K] (db) instead of being added by
  // atomics; head_backward_combine_kernel adds the workgroups in index order
  float* mine = ordered ? ordered + (size_t)blockIdx.x * (2 * K + 4) : nullptr;
  for (int pass = 0; pass < 2; ++pass) {
    float* out = pass == 0 ? dW : dbt;
    if (!out) continue;  // uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = lane * 4 + 256 * j;
      if (k < K) *reinterpret_cast<float4*>(&red[wave][k]) = pass == 0 ? gw[j] : gb[j];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
      const float t4 = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
      if (mine) mine[pass * K + k] = t4;
      else atomicAdd(&out[k], t4);
    }
    __syncthreads();
  }
  const float t = block_sum(sv, sh);
  if (threadIdx.x == 0 && db) {
    if (mine) mine[2 * K] = t;
    else atomicAdd(db, t);
  }
  if (amax) amax_commit(amax, amx);
}

// 64 outputs per workgroup; wave w adds the partials of workgroups w, w + 4, ... in increasing order (eight loads in flight), the four sums are
// added in wave order: a fixed order
__global__ __launch_bounds__(256) void head_backward_combine_kernel(const float* ordered, int blocks, int K, float* dW, float* db, float* dbt) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;  // 0..K) dW, K..2K) dbt, 2K db
  const size_t stride = (size_t)(2 * K + 4);
  float s = 0.f;
  if (i <= 2 * K) {
    int b = w;
    for (; b + 28 < blocks; b += 32) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ordered[(size_t)(b + 4 * k) * stride + i];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; b < blocks; b += 4) s += ordered[(size_t)b * stride + i];
  }
  part[w][c] = s;
  __syncthreads();
  if (w != 0 || i > 2 * K) return;
  float* out = i < K ? (dW ? dW + i : nullptr) : i < 2 * K ? (dbt ? dbt + (i - K) : nullptr) : db;
  if (out) *out += ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];  // (the outputs are accumulated into, as the atomics form does)
}

__global__ __launch_bounds__(256) void grad_penalty_kernel(const float* g, int ld, int dim, int M, float coef, float* G, unsigned short* G16, int planes16, float* stats, unsigned* amax) {
  __shared__ float sh[4];
  const int lane = threadIdx.x & 63;
  float sp = 0.f, amx = 0.f;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < M; r += gridDim.x * 4) {
    float sq = 0.f;
    for (int c = lane; c < dim; c += 64) { float x = g[(size_t)r * ld + c]; sq += x * x; }
    sq = wave_sum(sq);
    float n = sqrtf(sq + 1e-8f);  // add_agent.py:176
    float f = coef * 2.f * (n - 1.f) / n / (float)M;
    for (int c = lane; c < ld; c += 64) {
      const float o = c < dim ? f * g[(size_t)r * ld + c] : 0.f;
      if (G) G[(size_t)r * ld + c] = o;
      amx = fmaxf(amx, fabsf(o));
      if (G16) {
        if (planes16 == ADDHIP_STORE_BF16X3) addhip_planes::store1(G16 + 3 * (size_t)r * ld, c, o);
        else G16[(size_t)r * ld + c] = bf16_rne(o);
      }
    }
    if (lane == 0) sp += (n - 1.f) * (n - 1.f);
  }
  float t = block_sum(sp, sh);
  if (threadIdx.x == 0) atomicAdd(&stats[0], t);
  if (amax) amax_commit(amax, amx);
}

__global__ __launch_bounds__(256) void weighted_col_sum_kernel(const float* v, const float* X, int ld, int K, long long rows, float* out, float scale,
                                                               long long rows_per_block) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s = 0.f;
  if (c < K)
    for (long long r = r0 + w; r < r1; r += 4) s += v[r] * X[r * ld + c];
  part[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && c < K) atomicAdd(&out[c], scale * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]));
}

__global__ __launch_bounds__(256) void l2_grad_kernel(const float* w, float* grad, long long n, float coef, float* sumsq) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float x = w[i];
    grad[i] += coef * x;
    s += x * x;
  }
  float t = block_sum(s, sh);
  if (threadIdx.x == 0 && sumsq) atomicAdd(sumsq, t);
}

// torch.optim.AdamW single-tensor update order (weight decay, lerp, addcmul, sqrt/bias2 + eps, addcdiv); one definition for
// addhip_adamw and addhip_optimizer_step, so that the two agree bit for bit
__device__ __forceinline__ void adamw_one(float& p, float gi, float& m, float& v, float lr, float b1, float b2, float eps, float wd, float step_size,
                                          float bc2_sqrt) {
  float pi = p * (1.f - lr * wd);
  float mi = m + (1.f - b1) * (gi - m);
  float vi = v * b2 + (1.f - b2) * gi * gi;
  float denom = sqrtf(vi) / bc2_sqrt + eps;
  p = pi - step_size * (mi / denom);
  m = mi;
  v = vi;
}
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd,
                             float step_size, float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    adamw_one(pi, g[i], mi, vi, lr, b1, b2, eps, wd, step_size, bc2_sqrt);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

// torch.optim.SGD single-tensor update with momentum (mp_optimizer.py:33-36: momentum 0.9, dampening 0, no nesterov):
// g += wd p;  buf = first ? g : mu buf + g;  p -= lr buf
__device__ __forceinline__ void sgd_one(float& p, float gi, float& buf, float lr, float mu, float wd, int first) {
  if (wd != 0.f) gi = gi + wd * p;
  const float b = first ? gi : buf * mu + gi;
  buf = b;
  p = p - lr * b;
}
__global__ void sgd_kernel(float* p, const float* g, float* buf, long long n, float lr, float mu, float wd, int first) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float pi = p[i], bi = buf[i];
    sgd_one(pi, g[i], bi, lr, mu, wd, first);
    buf[i] = bi;
    p[i] = pi;
  }
}

// MPOptimizer.step as ONE pass over the flat buffers (addhip_optimizer_step): the AdamW / SGD update above element for element, plus --
// in the same read of the gradient and write of the parameter -- the bf16 weight shadow (round to nearest even) and the zero_grad of the
// NEXT step (mp_optimizer.py:14-16), so that an optimiser step ends with one launch instead of optimiser + shadow conversion + memset
__device__ __forceinline__ unsigned short opt_bf16(float v) {
  const unsigned u = __float_as_uint(v);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <bool SGD>
__global__ __launch_bounds__(256) void optimizer_step_kernel(addhip_optimizer_t o, float step_size, float bc2_sqrt) {
  const long long n4 = o.count >> 2;
  float4* p4 = reinterpret_cast<float4*>(o.param);
  float4* g4 = reinterpret_cast<float4*>(o.grad);
  float4* m4 = reinterpret_cast<float4*>(o.state1);
  float4* v4 = reinterpret_cast<float4*>(o.state2);
  auto one = [&](float& p, float g, float& m, float& v) {
    if (SGD) sgd_one(p, g, m, o.lr, o.beta1, o.weight_decay, o.step == 1 ? 1 : 0);
    else adamw_one(p, g, m, v, o.lr, o.beta1, o.beta2, o.eps, o.weight_decay, step_size, bc2_sqrt);
  };
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 p = p4[i], m = m4[i], v = SGD ? make_float4(0.f, 0.f, 0.f, 0.f) : v4[i];
    const float4 g = g4[i];
    one(p.x, g.x, m.x, v.x); one(p.y, g.y, m.y, v.y); one(p.z, g.z, m.z, v.z); one(p.w, g.w, m.w, v.w);
    p4[i] = p;
    m4[i] = m;
    if (!SGD) v4[i] = v;
    if (o.param16) {
      if (o.param16_planes == ADDHIP_STORE_BF16X3) addhip_planes::store4_flat(o.param16, 4 * i, p);
      else reinterpret_cast<uint2*>(o.param16)[i] = make_uint2((unsigned)opt_bf16(p.x) | ((unsigned)opt_bf16(p.y) << 16), (unsigned)opt_bf16(p.z) | ((unsigned)opt_bf16(p.w) << 16));
    }
    if (o.zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (o.count & 3)) {  // (count % 4 elements past the last float4)
    const long long i = (n4 << 2) + threadIdx.x;
    float v = SGD ? 0.f : o.state2[i];
    one(o.param[i], o.grad[i], o.state1[i], v);
    if (!SGD) o.state2[i] = v;
    if (o.param16) o.param16[i] = opt_bf16(o.param[i]);
    if (o.zero_grad) o.grad[i] = 0.f;
  }
}

// torch.nn.utils.clip_grad_norm_ (mp_optimizer.py:45-46): total 2-norm over the whole flat gradient, then
// grad *= min(1, max_norm / (norm + 1e-6)).  scratch[0] receives the sum of squares (double), scratch[1] the norm (stats).
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* g, long long n, double* scratch) {
  __shared__ double shd[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x = g[i];
    s += (double)x * x;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&scratch[0], shd[0] + shd[1] + shd[2] + shd[3]);
}
__global__ void grad_scale_kernel(float* g, long long n, float max_norm, const double* scratch, float* norm_out) {
  const float norm = (float)sqrt(scratch[0]);
  const float coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
  if (coef >= 1.0f) return;  // uniform
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) g[i] *= coef;
}

__global__ void return_tracker_fold_kernel(const float* ep, int T, float* state) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float episodes = state[0], mret = state[1], mlen = state[2];
  for (int t = 0; t < T; ++t) {
    float k = ep[3 * t + 2];
    if (k > 0.f) {  // base_agent.py:606-617
      float new_ret = ep[3 * t] / k, new_len = ep[3 * t + 1] / k;
      float cnt = episodes + k;
      float w_new = k / cnt, w_old = episodes / cnt;
      mret = w_new * new_ret + w_old * mret;
      mlen = w_new * new_len + w_old * mlen;
      episodes = cnt;
    }
  }
  state[0] = episodes; state[1] = mret; state[2] = mlen;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int addhip_fill_normal(float* out, int64_t count, uint64_t seed, uint64_t stream_id, void* stream) {
  ADDHIP_REQUIRE(out && count > 0, "fill_normal: bad arguments");
  ADDHIP_RECORDABLE(addhip_fill_normal, out, count, seed, stream_id);
  hipLaunchKernelGGL(fill_normal_kernel, dim3(elem_grid((count + 3) / 4)), dim3(256), 0, ST, out, (long long)count, seed, stream_id,
                     (const uint64_t*)nullptr);
  return addhip::check_launch("fill_normal_kernel");
}
extern "C" int addhip_fill_normal_at(float* out, int64_t count, uint64_t seed, uint64_t stream_id, const uint64_t* stream_base, void* stream) {
  ADDHIP_REQUIRE(out && count > 0 && stream_base, "fill_normal_at: bad arguments");
  ADDHIP_RECORDABLE(addhip_fill_normal_at, out, count, seed, stream_id, stream_base);
  hipLaunchKernelGGL(fill_normal_kernel, dim3(elem_grid((count + 3) / 4)), dim3(256), 0, ST, out, (long long)count, seed, stream_id, stream_base);
  return addhip::check_launch("fill_normal_kernel");
}
extern "C" int addhip_fill_uniform(float* out, int64_t count, uint64_t seed, uint64_t stream_id, void* stream) {
  ADDHIP_REQUIRE(out && count > 0, "fill_uniform: bad arguments");
  ADDHIP_RECORDABLE(addhip_fill_uniform, out, count, seed, stream_id);
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(elem_grid((count + 3) / 4)), dim3(256), 0, ST, out, (long long)count, seed, stream_id,
                     (const uint64_t*)nullptr);
  return addhip::check_launch("fill_uniform_kernel");
}
extern "C" int addhip_fill_uniform_at(float* out, int64_t count, uint64_t seed, uint64_t stream_id, const uint64_t* stream_base, void* stream) {
  ADDHIP_REQUIRE(out && count > 0 && stream_base, "fill_uniform_at: bad arguments");
  ADDHIP_RECORDABLE(addhip_fill_uniform_at, out, count, seed, stream_id, stream_base);
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(elem_grid((count + 3) / 4)), dim3(256), 0, ST, out, (long long)count, seed, stream_id, stream_base);
  return addhip::check_launch("fill_uniform_kernel");
}

extern "C" int addhip_dist_refresh(const float* logstd, float* dist, void* stream) {
  ADDHIP_REQUIRE(logstd && dist, "dist_refresh: null argument");
  ADDHIP_RECORDABLE(addhip_dist_refresh, logstd, dist);
  hipLaunchKernelGGL(dist_refresh_kernel, dim3(1), dim3(64), 0, ST, logstd, dist);
  return addhip::check_launch("dist_refresh_kernel");
}

extern "C" int addhip_actor_sample(const float* mean, int32_t ld_mean, const float* noise, float stdv, float logp_const, const float* dist, const float* logstd_rows, const float* a_mean,
                                   const float* a_std, int32_t num_envs, int32_t deterministic, const float* explore_u, float exp_prob, float* action,
                                   float* a_logp, float* rand_mask, void* stream) {
  ADDHIP_REQUIRE(mean && a_mean && a_std && action && a_logp && rand_mask && num_envs > 0, "actor_sample: bad arguments");
  ADDHIP_REQUIRE(deterministic || noise, "actor_sample: noise missing");
  ADDHIP_RECORDABLE(addhip_actor_sample, mean, ld_mean, noise, stdv, logp_const, dist, logstd_rows, a_mean, a_std, num_envs, deterministic, explore_u, exp_prob, action, a_logp, rand_mask);
  hipLaunchKernelGGL(actor_sample_kernel, dim3(row_grid(num_envs)), dim3(256), 0, ST, mean, ld_mean, noise, stdv, logp_const, dist, logstd_rows, a_mean, a_std,
                     num_envs, deterministic, explore_u, exp_prob, action, a_logp, rand_mask);
  return addhip::check_launch("actor_sample_kernel");
}

extern "C" int addhip_disc_prep(const float* disc_obs, const float* disc_demo, int32_t stride, int32_t dim, int64_t rows, const float* mean_abs,
                                 float min_diff, float* norm_diff, const int32_t* motion_id, const float* motion_time, const addhip_sampler_t* s,
                                 int32_t num_clips, float* abs_sum, void* stream) {
  ADDHIP_REQUIRE(disc_obs && disc_demo && mean_abs && rows > 0 && dim <= stride && stride <= 256, "disc_prep: bad arguments (stride <= 256)");
  ADDHIP_RECORDABLE(addhip_disc_prep, disc_obs, disc_demo, stride, dim, rows, mean_abs, min_diff, norm_diff, motion_id, motion_time, s, num_clips, abs_sum);
  addhip_sampler_t ss;
  memset(&ss, 0, sizeof(ss));
  int cells = 0;
  if (motion_id) {
    ADDHIP_REQUIRE(s && motion_time && s->err_sum && s->err_cnt && s->seg_size && num_clips > 0, "disc_prep: sampler state missing");
    ss = *s;
    cells = num_clips * s->num_segments;
    ADDHIP_REQUIRE(cells <= 8192, "disc_prep: more than 8192 (clip,segment) cells");
  }
  size_t shmem = sizeof(float) * (2 * (size_t)cells + 4 * (size_t)stride);
  hipLaunchKernelGGL(disc_prep_kernel, dim3(row_grid(rows) < 1024 ? row_grid(rows) : 1024), dim3(256), shmem, ST, disc_obs, disc_demo, stride, dim,
                     (long long)rows, mean_abs, min_diff, norm_diff, motion_id, motion_time, ss, cells, abs_sum);
  return addhip::check_launch("disc_prep_kernel");
}

extern "C" int addhip_sampler_update(const addhip_sampler_t* s, int32_t num_clips, void* stream) {
  ADDHIP_REQUIRE(s && s->errors && s->err_sum && s->err_cnt && num_clips > 0, "sampler_update: bad arguments");
  ADDHIP_RECORDABLE(addhip_sampler_update, s, num_clips);
  int cells = num_clips * s->num_segments;
  hipLaunchKernelGGL(sampler_update_kernel, dim3((cells + 255) / 256), dim3(256), 0, ST, *s, cells);
  return addhip::check_launch("sampler_update_kernel");
}

extern "C" int addhip_disc_reward(const float* logits, float* reward_inout, int64_t count, float scale, float task_w, float disc_w, float* stats,
                                  void* stream) {
  ADDHIP_REQUIRE(logits && reward_inout && count > 0, "disc_reward: bad arguments");
  ADDHIP_RECORDABLE(addhip_disc_reward, logits, reward_inout, count, scale, task_w, disc_w, stats);
  hipLaunchKernelGGL(disc_reward_kernel, dim3(elem_grid(count) < 1024 ? elem_grid(count) : 1024), dim3(256), 0, ST, logits, reward_inout,
                     (long long)count, scale, task_w, disc_w, stats);
  return addhip::check_launch("disc_reward_kernel");
}

extern "C" int addhip_head_gemv(const float* H, int32_t ld, int32_t K, int64_t rows, const float* w, const float* b, float* out, void* stream) {
  ADDHIP_REQUIRE(H && w && b && out && rows > 0 && K % 4 == 0 && ld % 4 == 0, "head_gemv: bad arguments");
  ADDHIP_RECORDABLE(addhip_head_gemv, H, ld, K, rows, w, b, out);
  hipLaunchKernelGGL(head_gemv_kernel, dim3(row_grid(rows)), dim3(256), 0, ST, H, ld, K, (long long)rows, w, b, out);
  return addhip::check_launch("head_gemv_kernel");
}

extern "C" int addhip_td_lambda_adv(const float* reward, const float* next_vals, const float* timeout_vals, const float* vals, const int32_t* done,
                                    const float* rand_mask, int32_t T, int32_t N, float discount, float td_lambda, float succ_val, float fail_val,
                                    float adv_clip, float* tar_val, float* adv, float* scratch, float* stats_out, void* stream) {
  ADDHIP_REQUIRE(reward && next_vals && vals && done && rand_mask && tar_val && adv && scratch && stats_out && T > 0 && N > 0,
                 "td_lambda_adv: bad arguments");
  ADDHIP_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7u) == 0, "td_lambda_adv: scratch must be 8-byte aligned");
  ADDHIP_RECORDABLE(addhip_td_lambda_adv, reward, next_vals, timeout_vals, vals, done, rand_mask, T, N, discount, td_lambda, succ_val, fail_val, adv_clip, tar_val, adv, scratch, stats_out);
  int blocks = (N + 255) / 256;
  ADDHIP_REQUIRE(blocks <= 1024, "td_lambda_adv: at most 262144 envs per rank");
  double* partial = reinterpret_cast<double*>(scratch);
  hipLaunchKernelGGL(td_lambda_kernel, dim3(blocks), dim3(256), 0, ST, reward, next_vals, timeout_vals, vals, done, rand_mask, T, N, discount, td_lambda,
                     succ_val, fail_val, tar_val, adv, partial);
  if (int rc = addhip::check_launch("td_lambda_kernel")) return rc;
  hipLaunchKernelGGL(adv_stats_kernel, dim3(1), dim3(64), 0, ST, partial, blocks, stats_out);
  if (int rc = addhip::check_launch("adv_stats_kernel")) return rc;
  hipLaunchKernelGGL(adv_norm_kernel, dim3(elem_grid((long long)T * N)), dim3(256), 0, ST, adv, (long long)T * N, stats_out, adv_clip);
  return addhip::check_launch("adv_norm_kernel");
}

extern "C" int addhip_norm_accum(const float* X, int64_t rows, int32_t dim, int32_t ld, float* sum, float* sumsq, void* stream) {
  ADDHIP_REQUIRE(X && sum && rows > 0 && dim > 0 && ld >= dim, "norm_accum: bad arguments");
  ADDHIP_RECORDABLE(addhip_norm_accum, X, rows, dim, ld, sum, sumsq);
  int strips = (dim + 63) / 64;
  int ysplit = 1;
  while (strips * ysplit < 512 && rows / (ysplit * 2) >= 64) ysplit *= 2;
  long long rpb = (rows + ysplit - 1) / ysplit;
  hipLaunchKernelGGL(norm_accum_kernel, dim3(strips, ysplit), dim3(256), 0, ST, X, (long long)rows, dim, ld, sum, sumsq, rpb);
  return addhip::check_launch("norm_accum_kernel");
}

extern "C" int addhip_norm_merge(float* mean, float* stdv, float* mean_sq, int64_t* count, float* sum, float* sumsq, int64_t new_count, int32_t dim,
                                 float min_var, int32_t first, void* stream) {
  ADDHIP_REQUIRE(mean && stdv && mean_sq && count && sum && sumsq && dim > 0 && dim <= 1024, "norm_merge: bad arguments (dim <= 1024)");
  ADDHIP_RECORDABLE(addhip_norm_merge, mean, stdv, mean_sq, count, sum, sumsq, new_count, dim, min_var, first);
  hipLaunchKernelGGL(norm_merge_kernel, dim3(1), dim3(1024), 0, ST, mean, stdv, mean_sq, (long long*)count, sum, sumsq, (long long)new_count, dim,
                     min_var, first);
  return addhip::check_launch("norm_merge_kernel");
}

extern "C" int addhip_diffnorm_merge(float* mean_abs, int64_t* count, float* abs_sum, int64_t new_count, int32_t dim, void* stream) {
  ADDHIP_REQUIRE(mean_abs && count && abs_sum && new_count > 0 && dim > 0 && dim <= 1024, "diffnorm_merge: bad arguments");
  ADDHIP_RECORDABLE(addhip_diffnorm_merge, mean_abs, count, abs_sum, new_count, dim);
  hipLaunchKernelGGL(diffnorm_merge_kernel, dim3(1), dim3(1024), 0, ST, mean_abs, (long long*)count, abs_sum, (long long)new_count, dim);
  return addhip::check_launch("diffnorm_merge_kernel");
}

extern "C" int addhip_amax_f32(const float* x, int64_t count, uint32_t* slots, void* stream) {
  ADDHIP_REQUIRE(x && slots && count > 0 && aligned16(x), "amax_f32: bad arguments (16-byte aligned input)");
  ADDHIP_RECORDABLE(addhip_amax_f32, x, count, slots);
  long long blocks = ((count >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks;
  hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, ST, x, (long long)count, slots);
  return addhip::check_launch("amax_kernel");
}

extern "C" int addhip_gather_minibatch(const addhip_gather_t* g, void* stream) {
  ADDHIP_REQUIRE(g && g->idx && g->count > 0, "gather: bad arguments");
  ADDHIP_REQUIRE(g->obs && g->obs_mean && g->obs_std && g->action && g->a_mean && g->a_std && g->a_logp && g->adv && g->tar_val && g->rand_mask &&
                     g->disc_obs && g->disc_demo && g->mean_abs, "gather: source pointers missing");
  ADDHIP_REQUIRE(g->norm_obs && g->norm_action && g->o_logp && g->o_adv && g->o_tar_val && g->o_mask && g->norm_diff, "gather: output pointers missing");
  ADDHIP_REQUIRE(g->planes16 == 0 || g->planes16 == ADDHIP_STORE_BF16 ||
                     (g->planes16 == ADDHIP_STORE_BF16X3 && g->obs_stride % 8 == 0 && g->disc_stride % 8 == 0 && aligned16(g->norm_obs16) && aligned16(g->norm_diff16)),
                 "gather: planes16 is an ADDHIP_STORE_* format (plane storage: row strides %% 8 == 0, 16-byte aligned buffers)");
  ADDHIP_RECORDABLE(addhip_gather_minibatch, g);
  hipLaunchKernelGGL(gather_kernel, dim3(row_grid(g->count)), dim3(256), 0, ST, *g);
  return addhip::check_launch("gather_kernel");
}

extern "C" int addhip_count_mask(const float* rand_mask, int32_t M, float* out, void* stream) {
  ADDHIP_REQUIRE(rand_mask && out && M > 0, "count_mask: bad arguments");
  ADDHIP_RECORDABLE(addhip_count_mask, rand_mask, M, out);
  ADDHIP_HIP(hipMemsetAsync(out, 0, sizeof(float), ST));
  hipLaunchKernelGGL(count_mask_kernel, dim3(elem_grid(M) < 64 ? elem_grid(M) : 64), dim3(256), 0, ST, rand_mask, M, out);
  return addhip::check_launch("count_mask_kernel");
}

extern "C" int addhip_actor_loss(const float* mean, const float* norm_action, const float* old_logp, const float* adv, const float* rand_mask, int32_t M,
                                 float stdv, float logp_const, const float* dist, float clip_ratio, float bound_weight, float reg_weight, float loss_scale,
                                 const float* n_valid, float* d_mean, float* g_logstd, float* stats, int32_t ld_mean, const float* logstd_rows, float entropy_weight,
                                 void* stream) {
  ADDHIP_REQUIRE(mean && norm_action && old_logp && adv && rand_mask && n_valid && d_mean && stats && M > 0, "actor_loss: bad arguments");
  ADDHIP_REQUIRE(!dist || g_logstd, "actor_loss: a trainable log-std (dist) needs g_logstd");
  ADDHIP_REQUIRE(ld_mean == 32 || (ld_mean == 64 && logstd_rows), "actor_loss: ld_mean is 32, or 64 with the per-sample log-std columns (logstd_rows)");
  ADDHIP_RECORDABLE(addhip_actor_loss, mean, norm_action, old_logp, adv, rand_mask, M, stdv, logp_const, dist, clip_ratio, bound_weight, reg_weight, loss_scale, n_valid, d_mean, g_logstd, stats,
                    ld_mean, logstd_rows, entropy_weight);
  hipLaunchKernelGGL(actor_loss_kernel, dim3(row_grid(M) < 256 ? row_grid(M) : 256), dim3(256), 0, ST, mean, norm_action, old_logp, adv, rand_mask, M,
                     stdv, logp_const, dist, clip_ratio, bound_weight, reg_weight, loss_scale, n_valid, d_mean, g_logstd, stats, ld_mean, logstd_rows, entropy_weight);
  return addhip::check_launch("actor_loss_kernel");
}

extern "C" int addhip_critic_head(const float* H, int32_t ld, int32_t K, int32_t M, const float* w, const float* b, const float* tar, float loss_scale,
                                  float* dZ, float* dv_out, float* stats, void* stream) {
  ADDHIP_REQUIRE(H && w && b && tar && dv_out && stats && M > 0 && K % 4 == 0 && ld % 4 == 0, "critic_head: bad arguments");
  ADDHIP_RECORDABLE(addhip_critic_head, H, ld, K, M, w, b, tar, loss_scale, dZ, dv_out, stats);
  hipLaunchKernelGGL(critic_head_kernel, dim3(row_grid(M) < 256 ? row_grid(M) : 256), dim3(256), 0, ST, H, ld, K, M, w, b, tar, loss_scale, dZ,
                     dv_out, stats);
  return addhip::check_launch("critic_head_kernel");
}

extern "C" int addhip_disc_head(const float* H, int32_t ld, int32_t K, int32_t M, const float* h_pos, const float* w, const float* b, float loss_scale,
                                float* dlogit, float* dlogit_pos, float* stats, void* stream) {
  ADDHIP_REQUIRE(H && h_pos && w && b && dlogit && dlogit_pos && stats && M > 0 && K % 4 == 0 && ld % 4 == 0, "disc_head: bad arguments");
  ADDHIP_RECORDABLE(addhip_disc_head, H, ld, K, M, h_pos, w, b, loss_scale, dlogit, dlogit_pos, stats);
  hipLaunchKernelGGL(disc_head_kernel, dim3(row_grid(M + 1) < 256 ? row_grid(M + 1) : 256), dim3(256), 0, ST, H, ld, K, M, h_pos, w, b, loss_scale,
                     dlogit, dlogit_pos, stats);
  return addhip::check_launch("disc_head_kernel");
}

extern "C" int addhip_outer_mask(const float* v, const float* w, const float* H, int32_t ld, int32_t K, int64_t rows, float* out, void* stream) {
  ADDHIP_REQUIRE(v && w && H && out && rows > 0 && K % 4 == 0 && ld % 4 == 0, "outer_mask: bad arguments");
  ADDHIP_RECORDABLE(addhip_outer_mask, v, w, H, ld, K, rows, out);
  hipLaunchKernelGGL(outer_mask_kernel, dim3(elem_grid(rows * (K / 4))), dim3(256), 0, ST, v, w, H, ld, K, (long long)rows, out, (unsigned short*)nullptr, 0, (unsigned*)nullptr);
  return addhip::check_launch("outer_mask_kernel");
}
#define ADDHIP_CHECK_PLANES16(who, buf, ldv)                                                                                                   \
  ADDHIP_REQUIRE(planes16 == 0 || planes16 == ADDHIP_STORE_BF16 || (planes16 == ADDHIP_STORE_BF16X3 && (!(buf) || ((ldv) % 8 == 0 && aligned16(buf)))), \
                 who ": planes16 is an ADDHIP_STORE_* format (plane storage: row length %% 8 == 0, 16-byte aligned buffer)")
extern "C" int addhip_bcast_mask(const float* w, const float* H, int32_t ld, int32_t K, int64_t rows, float* out, uint16_t* out16, int32_t planes16, uint32_t* amax,
                                 void* stream) {
  ADDHIP_REQUIRE(w && H && (out || out16) && rows > 0 && K % 4 == 0 && ld % 4 == 0, "bcast_mask: bad arguments");
  ADDHIP_CHECK_PLANES16("bcast_mask", out16, ld);
  ADDHIP_RECORDABLE(addhip_bcast_mask, w, H, ld, K, rows, out, out16, planes16, amax);
  hipLaunchKernelGGL(outer_mask_kernel, dim3(elem_grid(rows * (K / 4))), dim3(256), 0, ST, (const float*)nullptr, w, H, ld, K, (long long)rows, out, out16, planes16, amax);
  return addhip::check_launch("outer_mask_kernel(bcast)");
}

extern "C" int addhip_head_backward(const float* v, const float* w, const float* H, int32_t ld, int32_t K, int64_t rows, float* dZ, uint16_t* dZ16,
                                    int32_t planes16, float* dW_head, float* db_head, float* db_top, uint32_t* amax, float* ordered_scratch, void* stream) {
  ADDHIP_REQUIRE(v && w && H && rows > 0 && K > 0 && K <= 1024 && K % 4 == 0 && ld % 4 == 0 && ld >= K, "head_backward: bad arguments (K <= 1024)");
  ADDHIP_CHECK_PLANES16("head_backward", dZ16, ld);
  ADDHIP_RECORDABLE(addhip_head_backward, v, w, H, ld, K, rows, dZ, dZ16, planes16, dW_head, db_head, db_top, amax, ordered_scratch);
  const int grid = row_grid(rows) < 256 ? row_grid(rows) : 256;
  hipLaunchKernelGGL(head_backward_kernel, dim3(grid), dim3(256), 0, ST, v, w, H, ld, K, (long long)rows, dZ, dZ16, planes16, dW_head, db_head, db_top, amax,
                     ordered_scratch);
  if (ordered_scratch) hipLaunchKernelGGL(head_backward_combine_kernel, dim3((2 * K + 1 + 63) / 64), dim3(256), 0, ST, ordered_scratch, grid, K, dW_head, db_head, db_top);
  return addhip::check_launch("head_backward_kernel");
}

extern "C" int addhip_grad_penalty(const float* g, int32_t ld, int32_t dim, int32_t M, float coef, float* G, uint16_t* G16, int32_t planes16, float* stats, uint32_t* amax,
                                   void* stream) {
  ADDHIP_REQUIRE(g && (G || G16) && stats && M > 0 && dim <= ld, "grad_penalty: bad arguments");
  ADDHIP_CHECK_PLANES16("grad_penalty", G16, ld);
  ADDHIP_RECORDABLE(addhip_grad_penalty, g, ld, dim, M, coef, G, G16, planes16, stats, amax);
  hipLaunchKernelGGL(grad_penalty_kernel, dim3(row_grid(M) < 256 ? row_grid(M) : 256), dim3(256), 0, ST, g, ld, dim, M, coef, G, G16, planes16, stats, amax);
  return addhip::check_launch("grad_penalty_kernel");
}

extern "C" int addhip_weighted_col_sum(const float* v, const float* X, int32_t ld, int32_t K, int64_t rows, float* out, float scale, int32_t accumulate,
                                       void* stream) {
  ADDHIP_REQUIRE(v && X && out && rows > 0 && K > 0 && ld >= K, "weighted_col_sum: bad arguments");
  ADDHIP_RECORDABLE(addhip_weighted_col_sum, v, X, ld, K, rows, out, scale, accumulate);
  if (!accumulate) ADDHIP_HIP(hipMemsetAsync(out, 0, sizeof(float) * K, ST));
  int strips = (K + 63) / 64;
  int ysplit = 1;
  while (strips * ysplit < 512 && rows / (ysplit * 2) >= 64) ysplit *= 2;
  long long rpb = (rows + ysplit - 1) / ysplit;
  hipLaunchKernelGGL(weighted_col_sum_kernel, dim3(strips, ysplit), dim3(256), 0, ST, v, X, ld, K, (long long)rows, out, scale, rpb);
  return addhip::check_launch("weighted_col_sum_kernel");
}

extern "C" int addhip_l2_grad(const float* w, float* grad, int64_t count, float coef, float* sumsq_out, void* stream) {
  ADDHIP_REQUIRE(w && grad && count > 0, "l2_grad: bad arguments");
  ADDHIP_RECORDABLE(addhip_l2_grad, w, grad, count, coef, sumsq_out);
  hipLaunchKernelGGL(l2_grad_kernel, dim3(elem_grid(count) < 512 ? elem_grid(count) : 512), dim3(256), 0, ST, w, grad, (long long)count, coef, sumsq_out);
  return addhip::check_launch("l2_grad_kernel");
}

extern "C" int addhip_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int32_t step, void* stream) {
  ADDHIP_REQUIRE(param && grad && exp_avg && exp_avg_sq && count > 0 && step >= 1, "adamw: bad arguments");
  ADDHIP_RECORDABLE(addhip_adamw, param, grad, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps, weight_decay, step);
  // bias corrections in double like torch's python-scalar path (torch/optim/adamw.py single-tensor)
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1);
  float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adamw_kernel, dim3(elem_grid(count)), dim3(256), 0, ST, param, grad, exp_avg, exp_avg_sq, (long long)count, lr, beta1, beta2, eps,
                     weight_decay, step_size, bc2_sqrt);
  return addhip::check_launch("adamw_kernel");
}

extern "C" int addhip_sgd(float* param, const float* grad, float* momentum_buf, int64_t count, float lr, float momentum, float weight_decay,
                          int32_t step, void* stream) {
  ADDHIP_REQUIRE(param && grad && momentum_buf && count > 0 && step >= 1, "sgd: bad arguments");
  ADDHIP_RECORDABLE(addhip_sgd, param, grad, momentum_buf, count, lr, momentum, weight_decay, step);
  hipLaunchKernelGGL(sgd_kernel, dim3(elem_grid(count)), dim3(256), 0, ST, param, grad, momentum_buf, (long long)count, lr, momentum, weight_decay,
                     step == 1 ? 1 : 0);
  return addhip::check_launch("sgd_kernel");
}

extern "C" int addhip_optimizer_step(const addhip_optimizer_t* op, void* stream) {
  ADDHIP_REQUIRE(op, "optimizer_step: null descriptor");
  ADDHIP_RECORDABLE(addhip_optimizer_step, op);
  const addhip_optimizer_t o = *op;
  ADDHIP_REQUIRE(o.type == ADDHIP_OPT_ADAMW || o.type == ADDHIP_OPT_SGD, "optimizer_step: type must be ADDHIP_OPT_ADAMW or ADDHIP_OPT_SGD");
  ADDHIP_REQUIRE(o.param && o.grad && o.state1 && (o.state2 || o.type == ADDHIP_OPT_SGD) && o.count > 0 && o.step >= 1, "optimizer_step: bad arguments");
  ADDHIP_REQUIRE(aligned16(o.param) && aligned16(o.grad) && aligned16(o.state1) && (o.type == ADDHIP_OPT_SGD || aligned16(o.state2)) &&
                     (reinterpret_cast<uintptr_t>(o.param16) & 7u) == 0, "optimizer_step: buffers must be 16-byte aligned (the bf16 shadow: 8)");
  ADDHIP_REQUIRE(o.param16_planes == 0 || o.param16_planes == ADDHIP_STORE_BF16 || (o.param16_planes == ADDHIP_STORE_BF16X3 && o.count % 8 == 0 && aligned16(o.param16)),
                 "optimizer_step: param16_planes is an ADDHIP_STORE_* format (plane storage: count %% 8 == 0, 16-byte aligned shadow)");
  long long blocks = ((o.count >> 2) + 255) / 256;
  blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
  if (o.type == ADDHIP_OPT_SGD) {
    hipLaunchKernelGGL(optimizer_step_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ST, o, 0.f, 1.f);
  } else {
    const double bc1 = 1.0 - pow((double)o.beta1, (double)o.step), bc2 = 1.0 - pow((double)o.beta2, (double)o.step);
    hipLaunchKernelGGL(optimizer_step_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ST, o, (float)((double)o.lr / bc1), (float)sqrt(bc2));
  }
  return addhip::check_launch("optimizer_step_kernel");
}

extern "C" int addhip_grad_clip(float* grad, int64_t count, float max_norm, float* scratch, float* norm_out, void* stream) {
  ADDHIP_REQUIRE(grad && scratch && count > 0 && max_norm > 0.0f, "grad_clip: bad arguments");
  ADDHIP_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7u) == 0, "grad_clip: scratch must be 8-byte aligned");
  ADDHIP_RECORDABLE(addhip_grad_clip, grad, count, max_norm, scratch, norm_out);
  double* sc = reinterpret_cast<double*>(scratch);
  ADDHIP_HIP(hipMemsetAsync(sc, 0, sizeof(double), ST));
  const int grid = elem_grid(count) < 1024 ? elem_grid(count) : 1024;
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(grid), dim3(256), 0, ST, grad, (long long)count, sc);
  if (int rc = addhip::check_launch("grad_sumsq_kernel")) return rc;
  hipLaunchKernelGGL(grad_scale_kernel, dim3(elem_grid(count)), dim3(256), 0, ST, grad, (long long)count, max_norm, sc, norm_out);
  return addhip::check_launch("grad_scale_kernel");
}

extern "C" int addhip_return_tracker_fold(const float* ep_stats, int32_t T, float* state, void* stream) {
  ADDHIP_REQUIRE(ep_stats && state && T > 0, "return_tracker_fold: bad arguments");
  ADDHIP_RECORDABLE(addhip_return_tracker_fold, ep_stats, T, state);
  hipLaunchKernelGGL(return_tracker_fold_kernel, dim3(1), dim3(64), 0, ST, ep_stats, T, state);
  return addhip::check_launch("return_tracker_fold_kernel");
}

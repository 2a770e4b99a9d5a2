// Per-env hot path of the rollout: reference-motion frame lookup, history ring, observation /
// discriminator-observation assembly, imitation reward, done flags, return tracker, masked reset.
//
// One wavefront (64 lanes) owns one env at a time.  All rows an env needs (simulator state,
// reference frame, 6 target frames, 2 older demo frames, 2 older history frames: 16 x 36 floats)
// are fetched with 16-byte loads -- 9 lanes per 144-byte row, 7 rows per wave instruction --
// into a wave-private LDS tile; outputs are then produced output-major (lane = output column) so
// every global store is a contiguous run of the obs / disc rows.  Reductions (pose / velocity
// error sums, softmax over sampler segments) are wavefront shuffles.  HBM-bound by construction:
// 4.6 KB of algorithmic traffic per env-step.
//
// Reference functions restated here: see include/addhip.h at each entry point.
#include "common.h"
#include "quat.cuh"

using namespace addhip;

namespace {

constexpr int PW = ADDHIP_POSE_W;      // 36 floats per row
constexpr int ROWS = 16;
constexpr int WAVES = 4;
constexpr int R_SIM = 0, R_SIMV = 1, R_REF = 2, R_REFV = 3, R_TAR = 4, R_DEMO0 = 12, R_DEMO1 = 13, R_H0 = 14, R_H1 = 15;

// MotionLib.get_precomputed_motion_step index (anim/motion_lib.py:322-326): fp32 multiply by
// round(1/dt), truncate toward zero, clamp, add the clip offset.  Bit-exact by construction:
// explicit round-to-nearest multiply, no contraction.
__device__ __forceinline__ int step_index(const addhip_motion_t& m, int id, float t) {
  float f = __fmul_rn(t, m.dt_inv);
  long long fr = (long long)f;
  if (m.reference_compat) {
    long long hi = (long long)m.total_steps - 1;
    fr = fr < 0 ? 0 : (fr > hi ? hi : fr);
    long long idx = fr + (long long)m.clip_start[id];
    return (int)(idx > hi ? hi : idx);  // the reference would raise IndexError here
  }
  long long hi = (long long)m.clip_steps[id] - 1;
  fr = fr < 0 ? 0 : (fr > hi ? hi : fr);
  return (int)(fr + (long long)m.clip_start[id]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ Quat row_quat(const float* r) { return Quat{r[3], r[4], r[5], r[6]}; }

// one element of compute_add_obs (add_observation.py:653-717 with compute_char_obs :422-459 and
// compute_tar_obs :578-650; vel/phase observations are not part of this build's HIP path)
__device__ __forceinline__ float obs_elem(const addhip_task_t& t, const float (*rows)[PW], int o) {
  const int hc = t.root_height_obs ? 1 : 0;
  const int char_dim = hc + 6 + ADDHIP_NUM_DOF;
  const float* sim = rows[R_SIM];
  if (o < char_dim) {
    if (hc && o == 0) return sim[2];
    int c = o - hc;
    if (c < 6) {
      Quat q = row_quat(sim);
      if (!t.global_obs) q = quat_mul(heading_quat_inv(q), q);
      return tan_norm_elem(q, c);
    }
    return sim[7 + c - 6];
  }
  const int pw = hc ? 3 : 2;
  const int tw = pw + 6 + ADDHIP_NUM_DOF;
  int k = (o - char_dim) / tw;
  int c = (o - char_dim) - k * tw;
  const float* tar = rows[R_TAR + k];
  if (c < pw) {
    if (c == 2) return tar[2];  // add_observation.py:615-616 absolute height
    if (t.global_obs) return tar[c] - sim[c];
    const float* t0 = rows[R_TAR];
    Vec3 d{tar[0] - t0[0], tar[1] - t0[1], tar[2] - t0[2]};
    Vec3 r = quat_rotate(heading_quat_inv(row_quat(t0)), d);
    return c == 0 ? r.x : r.y;
  }
  c -= pw;
  if (c < 6) {
    Quat q = row_quat(tar);
    if (!t.global_obs) q = quat_mul(heading_quat_inv(row_quat(rows[R_TAR])), q);
    return tan_norm_elem(q, c);
  }
  return tar[7 + c - 6];
}

// one element of compute_disc_obs (add_observation.py:462-554) for history rows (r0,r1,r2) oldest..newest
__device__ __forceinline__ float disc_elem(const addhip_task_t& t, const float* r0, const float* r1, const float* r2, int o) {
  constexpr int sw = 3 + 6 + ADDHIP_NUM_DOF;  // 38
  int s = o / sw;
  int c = o - s * sw;
  const float* r = s == 0 ? r0 : (s == 1 ? r1 : r2);
  if (c < 3) return (!t.global_obs && c < 2) ? 0.0f : r[c];
  if (c < 9) return tan_norm_elem(row_quat(r), c - 3);
  return r[7 + c - 9];
}

struct RowSrc {
  const float* p[ROWS];
};

// which global row feeds LDS row r (fresh=1: state just (re)initialised from the clip)
__device__ __forceinline__ const float* row_ptr(const addhip_motion_t& m, const addhip_task_t& t, const addhip_env_t& e,
                                                int env, int r, int id, float tm, int head_old0, int head_old1, bool fresh) {
  if (r == R_REF || (fresh && r == R_SIM)) return m.pose + (size_t)step_index(m, id, tm) * PW;
  if (r == R_REFV || (fresh && r == R_SIMV)) return m.vel + (size_t)step_index(m, id, tm) * PW;
  if (r == R_SIM) return e.sim_pose + (size_t)env * PW;
  if (r == R_SIMV) return e.sim_vel + (size_t)env * PW;
  if (r >= R_TAR && r < R_TAR + ADDHIP_MAX_TAR_STEPS) {
    int k = r - R_TAR;
    if (k >= t.num_tar_steps) k = 0;
    return m.pose + (size_t)step_index(m, id, __fadd_rn(tm, t.tar_dt[k])) * PW;  // add_observation.py:214-215
  }
  if (r == R_DEMO0 || (fresh && r == R_H0)) return m.pose + (size_t)step_index(m, id, __fadd_rn(tm, t.demo_dt[0])) * PW;
  if (r == R_DEMO1 || (fresh && r == R_H1)) return m.pose + (size_t)step_index(m, id, __fadd_rn(tm, t.demo_dt[1])) * PW;
  if (r == R_H0) return e.hist + ((size_t)env * ADDHIP_HIST + head_old0) * PW;
  return e.hist + ((size_t)env * ADDHIP_HIST + head_old1) * PW;
}

__device__ __forceinline__ void stage_rows(float (*rows)[PW], const addhip_motion_t& m, const addhip_task_t& t,
                                           const addhip_env_t& e, int env, int id, float tm, int h0, int h1, bool fresh, int lane) {
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    int item = lane + 64 * p;
    if (item < ROWS * 9) {
      int r = item / 9, part = item - r * 9;
      const float4* src = reinterpret_cast<const float4*>(row_ptr(m, t, e, env, r, id, tm, h0, h1, fresh));
      float4 v = src[part];
      *reinterpret_cast<float4*>(&rows[r][part * 4]) = v;
    }
  }
}

__device__ __forceinline__ void emit_obs(const addhip_task_t& t, const float (*rows)[PW], int env, int lane,
                                         float* obs, float* obs2, float* disc, float* demo) {
  if (obs || obs2) {
    for (int o = lane; o < t.obs_stride; o += 64) {
      float v = o < t.obs_dim ? obs_elem(t, rows, o) : 0.0f;
      if (obs) obs[(size_t)env * t.obs_stride + o] = v;
      if (obs2) obs2[(size_t)env * t.obs_stride + o] = v;
    }
  }
  if (disc) {
    for (int o = lane; o < t.disc_stride; o += 64)
      disc[(size_t)env * t.disc_stride + o] = o < t.disc_dim ? disc_elem(t, rows[R_H0], rows[R_H1], rows[R_SIM], o) : 0.0f;
  }
  if (demo) {
    for (int o = lane; o < t.disc_stride; o += 64)
      demo[(size_t)env * t.disc_stride + o] = o < t.disc_dim ? disc_elem(t, rows[R_DEMO0], rows[R_DEMO1], rows[R_REF], o) : 0.0f;
  }
}

__global__ __launch_bounds__(64 * WAVES) void env_step_kernel(addhip_motion_t m, addhip_task_t t, addhip_env_t e,
                                                              addhip_step_out_t o, int head) {
  __shared__ __attribute__((aligned(16))) float lds[WAVES][ROWS][PW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float(*rows)[PW] = lds[w];
  const int groups = (e.num_envs + WAVES - 1) / WAVES;
  const int h0 = (head + 1) % ADDHIP_HIST, h1 = (head + 2) % ADDHIP_HIST;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int env = g * WAVES + w;
    const bool valid = env < e.num_envs;
    float time_new = 0.0f, tm = 0.0f;
    int id = 0;
    if (valid) {
      time_new = __fadd_rn(e.time[env], t.dt);            // env.py:155
      id = e.motion_id[env];
      tm = __fadd_rn(time_new, e.time_off[env]);          // add_observation.py:352-354
      stage_rows(rows, m, t, e, env, id, tm, h0, h1, false, lane);
    }
    __syncthreads();
    if (valid) {
      emit_obs(t, rows, env, lane, o.obs, o.obs_next_in, o.disc_obs, o.disc_demo);
      // history push (circular_buffer.py:17-20) and reference state (add_observation.py:163-174)
      if (lane < PW) {
        e.hist[((size_t)env * ADDHIP_HIST + head) * PW + lane] = rows[R_SIM][lane];
        if (e.ref_pose) e.ref_pose[(size_t)env * PW + lane] = rows[R_REF][lane];
        if (e.ref_vel) e.ref_vel[(size_t)env * PW + lane] = rows[R_REFV][lane];
      }
      // ---- reward (add_reward.py:103-177), joint weights are all 1 (add_reward.py:31-34)
      float pe = 0.0f, ve = 0.0f;
      if (lane < ADDHIP_NUM_DOF) {
        float d = rows[R_REF][7 + lane] - rows[R_SIM][7 + lane];
        pe = d * d;
        float dv = rows[R_REFV][6 + lane] - rows[R_SIMV][6 + lane];
        ve = dv * dv;
      }
      pe = wave_sum(pe);
      ve = wave_sum(ve);
      const float* sim = rows[R_SIM];
      const float* ref = rows[R_REF];
      const bool track_root = (t.num_tar_steps > 0) && t.global_obs;  // add_observation.py:349-350
      float dx = ref[0] - sim[0], dy = ref[1] - sim[1], dz = ref[2] - sim[2];
      float root_err_full = dx * dx + dy * dy + dz * dz;
      float rx = track_root ? dx : 0.0f, ry = track_root ? dy : 0.0f, rz = t.root_height_obs ? dz : 0.0f;
      float root_pos_err = rx * rx + ry * ry + rz * rz;
      Quat q_sim = row_quat(sim), q_ref = row_quat(ref);
      Vec3 v_sim{rows[R_SIMV][0], rows[R_SIMV][1], rows[R_SIMV][2]}, w_sim{rows[R_SIMV][3], rows[R_SIMV][4], rows[R_SIMV][5]};
      Vec3 v_ref{rows[R_REFV][0], rows[R_REFV][1], rows[R_REFV][2]}, w_ref{rows[R_REFV][3], rows[R_REFV][4], rows[R_REFV][5]};
      if (!track_root) {  // convert_to_local_root (add_reward.py:91-101)
        Quat hs = heading_quat_inv(q_sim), hr = heading_quat_inv(q_ref);
        v_sim = quat_rotate(hs, v_sim); w_sim = quat_rotate(hs, w_sim); q_sim = quat_mul(hs, q_sim);
        v_ref = quat_rotate(hr, v_ref); w_ref = quat_rotate(hr, w_ref); q_ref = quat_mul(hr, q_ref);
      }
      float rot_err = quat_diff_angle(q_sim, q_ref);
      rot_err *= rot_err;
      float vx = v_ref.x - v_sim.x, vy = v_ref.y - v_sim.y, vz = v_ref.z - v_sim.z;
      float root_vel_err = vx * vx + vy * vy + vz * vz;
      float ax = w_ref.x - w_sim.x, ay = w_ref.y - w_sim.y, az = w_ref.z - w_sim.z;
      float root_ang_err = ax * ax + ay * ay + az * az;
      float r = t.pose_w * expf(-t.pose_scale * pe) + t.vel_w * expf(-t.vel_scale * ve) +
                t.root_pose_w * expf(-t.root_pose_scale * (root_pos_err + 0.1f * rot_err)) +
                t.root_vel_w * expf(-t.root_vel_scale * (root_vel_err + 0.1f * root_ang_err));
      // ---- done (add_done.py:96-147)
      if (lane == 0) {
        int done = ADDHIP_DONE_NULL;
        if (time_new >= t.max_episode_length) done = ADDHIP_DONE_TIME;
        if (tm >= m.clip_len[id] && m.clip_loop[id] != 1) done = ADDHIP_DONE_SUCC;
        if (t.enable_early_termination) {
          bool failed = e.contact ? (e.contact[env] != 0) : false;
          if (t.pose_termination) {
            bool pose_fail = (pe / (float)ADDHIP_NUM_DOF) > t.pose_termination_dist;
            if (track_root) pose_fail = pose_fail || (root_err_full > t.pose_termination_dist);
            failed = failed || pose_fail;
          }
          if (failed && time_new > 0.0f) done = ADDHIP_DONE_FAIL;
        }
        e.time[env] = time_new;
        e.done[env] = done;
        if (o.done) o.done[env] = done;
        if (o.reward) o.reward[env] = r;
        if (o.motion_id_rec) o.motion_id_rec[env] = id;
        if (o.motion_time_rec) o.motion_time_rec[env] = tm;
        if (e.ret_acc) {  // ReturnTracker.update (base_agent.py:596-621)
          float ra = e.ret_acc[env] + r;
          int la = e.len_acc[env] + 1;
          if (done != ADDHIP_DONE_NULL) {
            if (o.ep_stats) {
              atomicAdd(&o.ep_stats[0], ra);
              atomicAdd(&o.ep_stats[1], (float)la);
              atomicAdd(&o.ep_stats[2], 1.0f);
            }
            ra = 0.0f;
            la = 0;
          }
          e.ret_acc[env] = ra;
          e.len_acc[env] = la;
        }
      }
    }
    __syncthreads();
  }
}

// ---- reset phase 1: clip draw (MotionLib.sample_motions, motion_lib.py:35-39) + batch temperature
__global__ void reset_draw_kernel(addhip_motion_t m, addhip_env_t e, addhip_sampler_t s, const float* u_clip, int reset_all) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= e.num_envs) return;
  if (!reset_all && e.done[env] == ADDHIP_DONE_NULL) return;
  float u = u_clip[env];
  int id = 0;
  while (id < m.num_clips - 1 && !(u < s.clip_cdf[id])) ++id;
  e.motion_id[env] = id;
  if (s.temperature <= 0.0f) {
    float mx = 0.0f;  // errors are >= 0
    for (int k = 0; k < s.num_segments; ++k) mx = fmaxf(mx, s.errors[id * s.num_segments + k]);
    atomicMax(s.temp_bits, __float_as_uint(mx));
  }
}

// c10::div_floor_floating, the arithmetic behind `time // dt` (sampler.py:88)
__device__ __forceinline__ float floor_div_f32(float a, float b) {
  float mod = fmodf(a, b);
  float div = __fdiv_rn(__fsub_rn(a, mod), b);
  if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) div = __fsub_rn(div, 1.0f);
  if (div != 0.0f) {
    float fl = floorf(div);
    if (__fsub_rn(div, fl) > 0.5f) fl = __fadd_rn(fl, 1.0f);
    return fl;
  }
  return copysignf(0.0f, __fdiv_rn(a, b));
}

// ---- reset phase 2: start time, state from the clip, history refill, observations
__global__ __launch_bounds__(64 * WAVES) void env_reset_kernel(addhip_motion_t m, addhip_task_t t, addhip_env_t e, addhip_sampler_t s,
                                                               const float* u_seg, const float* u_jit, float* obs_out,
                                                               float* disc_out, float* demo_out, int reset_all, int head) {
  __shared__ __attribute__((aligned(16))) float lds[WAVES][ROWS][PW];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float(*rows)[PW] = lds[w];
  const int groups = (e.num_envs + WAVES - 1) / WAVES;
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int env = g * WAVES + w;
    const bool active = env < e.num_envs && (reset_all || e.done[env] != ADDHIP_DONE_NULL);
    int id = 0;
    float off = 0.0f;
    if (active) {
      id = e.motion_id[env];
      if (s.rand_reset) {
        // AdaptiveSegmentSampler.get_probs + multinomial by inverse CDF (sampler.py:57-80)
        const int S = s.num_segments;
        float temp = s.temperature > 0.0f ? s.temperature : __fadd_rn(__uint_as_float(*s.temp_bits), 1e-6f);
        float z = lane < S ? s.errors[id * S + lane] / temp : -INFINITY;
        float zmax = wave_max(z);
        float ez = lane < S ? expf(z - zmax) : 0.0f;
        float p = ez / wave_sum(ez);
        float cdf = p;  // inclusive scan over the first S lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          float up = __shfl_up(cdf, d, 64);
          if (lane >= d) cdf += up;
        }
        float u = u_seg[env];
        unsigned long long below = __ballot(lane < S && !(u < cdf));
        int seg = __popcll(below);
        if (seg > S - 1) seg = S - 1;
        float ss = s.seg_size[id];
        float tt = __fmul_rn((float)seg, ss);                 // sampler.py:81-82
        tt = __fadd_rn(tt, __fmul_rn(u_jit[env], ss));        // :84-85
        tt = __fmul_rn(floor_div_f32(tt, t.dt), t.dt);        // :88
        off = fmaxf(tt, s.min_start_time);                    // :91
      }
      // time_buf = 0 (env.py:161) -> motion time == offset
      stage_rows(rows, m, t, e, env, id, off, 0, 0, true, lane);
    }
    __syncthreads();
    if (active) {
      emit_obs(t, rows, env, lane, obs_out, nullptr, disc_out, demo_out);
      if (lane < PW) {
        // set_qpos / set_dofs_velocity payload (add_observation.py:314-331) straight into the simulator state
        e.sim_pose[(size_t)env * PW + lane] = rows[R_REF][lane];
        e.sim_vel[(size_t)env * PW + lane] = rows[R_REFV][lane];
        if (e.ref_pose) e.ref_pose[(size_t)env * PW + lane] = rows[R_REF][lane];
        if (e.ref_vel) e.ref_vel[(size_t)env * PW + lane] = rows[R_REFV][lane];
        // CircularBuffer.fill (circular_buffer.py:22-29): get_all() order = demo frames t-2dt, t-dt, t
        float* hb = e.hist + (size_t)env * ADDHIP_HIST * PW;
        hb[((head + 0) % ADDHIP_HIST) * PW + lane] = rows[R_DEMO0][lane];
        hb[((head + 1) % ADDHIP_HIST) * PW + lane] = rows[R_DEMO1][lane];
        hb[((head + 2) % ADDHIP_HIST) * PW + lane] = rows[R_REF][lane];
      }
      if (lane == 0) {
        e.time[env] = 0.0f;
        e.time_off[env] = off;
        e.done[env] = ADDHIP_DONE_NULL;  // add_done.py:92-93
      }
    }
    __syncthreads();
  }
}

__global__ void lookup_kernel(addhip_motion_t m, const int* ids, const float* times, int count, int* idx_out, float* pose_out, float* vel_out) {
  int q = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (q >= count) return;
  int idx = step_index(m, ids[q], times[q]);
  if (lane == 0 && idx_out) idx_out[q] = idx;
  if (lane < PW) {
    if (pose_out) pose_out[(size_t)q * PW + lane] = m.pose[(size_t)idx * PW + lane];
    if (vel_out) vel_out[(size_t)q * PW + lane] = m.vel[(size_t)idx * PW + lane];
  }
}

__global__ void kin_engine_step_kernel(float* sim_pose, float* sim_vel, const float* target, int tstride, int n, float lag, float dt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ADDHIP_NUM_DOF) return;
  int env = i / ADDHIP_NUM_DOF, j = i - env * ADDHIP_NUM_DOF;
  float q = sim_pose[(size_t)env * PW + 7 + j];
  float qn = __fadd_rn(q, __fmul_rn(lag, __fsub_rn(target[(size_t)env * tstride + j], q)));
  sim_vel[(size_t)env * PW + 6 + j] = __fdiv_rn(__fsub_rn(qn, q), dt);
  sim_pose[(size_t)env * PW + 7 + j] = qn;
}

int check_common(const addhip_motion_t* m, const addhip_task_t* t, const addhip_env_t* e) {
  ADDHIP_REQUIRE(m && t && e, "null struct");
  ADDHIP_REQUIRE(e->num_envs > 0, "num_envs must be > 0");
  ADDHIP_REQUIRE(m->pose && m->vel && m->clip_start && m->clip_len && m->clip_loop && m->clip_steps, "motion tables missing");
  ADDHIP_REQUIRE(m->total_steps > 0 && m->num_clips > 0, "empty motion library");
  ADDHIP_REQUIRE(t->num_tar_steps >= 0 && t->num_tar_steps <= ADDHIP_MAX_TAR_STEPS, "num_tar_steps out of range");
  ADDHIP_REQUIRE(t->demo_dt[ADDHIP_HIST - 1] == 0.0f, "demo_dt[last] must be 0 (newest demo frame == reference frame)");
  const int hc = t->root_height_obs ? 1 : 0;
  const int want = hc + 6 + ADDHIP_NUM_DOF + t->num_tar_steps * ((hc ? 3 : 2) + 6 + ADDHIP_NUM_DOF);
  ADDHIP_REQUIRE(t->obs_dim == want, "obs_dim %d does not match the task flags (expected %d)", t->obs_dim, want);
  ADDHIP_REQUIRE(t->disc_dim == ADDHIP_HIST * (9 + ADDHIP_NUM_DOF), "disc_dim must be %d", ADDHIP_HIST * (9 + ADDHIP_NUM_DOF));
  ADDHIP_REQUIRE(t->obs_stride >= t->obs_dim && t->disc_stride >= t->disc_dim, "strides smaller than dims");
  ADDHIP_REQUIRE(e->sim_pose && e->sim_vel && e->time && e->time_off && e->motion_id && e->hist && e->done, "env state pointers missing");
  ADDHIP_REQUIRE(aligned16(m->pose) && aligned16(m->vel) && aligned16(e->sim_pose) && aligned16(e->sim_vel) && aligned16(e->hist),
                 "row buffers must be 16-byte aligned");
  return 0;
}

inline int env_grid(int num_envs) {
  int groups = (num_envs + WAVES - 1) / WAVES;
  return groups < 2048 ? groups : 2048;
}

}  // namespace

extern "C" int addhip_env_step(const addhip_motion_t* m, const addhip_task_t* t, const addhip_env_t* e,
                               const addhip_step_out_t* o, int32_t head, void* stream) {
  if (int rc = check_common(m, t, e)) return rc;
  ADDHIP_REQUIRE(o, "null outputs");
  ADDHIP_REQUIRE(head >= 0 && head < ADDHIP_HIST, "head out of range");
  hipLaunchKernelGGL(env_step_kernel, dim3(env_grid(e->num_envs)), dim3(64 * WAVES), 0, (hipStream_t)stream, *m, *t, *e, *o, head);
  return check_launch("env_step_kernel");
}

extern "C" int addhip_env_reset(const addhip_motion_t* m, const addhip_task_t* t, const addhip_env_t* e,
                                const addhip_sampler_t* s, const float* u_clip, const float* u_seg, const float* u_jit,
                                float* obs_out, float* disc_obs_out, float* disc_demo_out, int32_t reset_all, int32_t head,
                                void* stream) {
  if (int rc = check_common(m, t, e)) return rc;
  ADDHIP_REQUIRE(s && s->errors && s->seg_size && s->clip_cdf && s->temp_bits, "sampler pointers missing");
  ADDHIP_REQUIRE(s->num_segments > 0 && s->num_segments <= 64, "num_segments must be in 1..64");
  ADDHIP_REQUIRE(u_clip && u_seg && u_jit, "uniform draws missing");
  ADDHIP_REQUIRE(head >= 0 && head < ADDHIP_HIST, "head out of range");
  hipStream_t st = (hipStream_t)stream;
  ADDHIP_HIP(hipMemsetAsync(s->temp_bits, 0, sizeof(uint32_t), st));
  hipLaunchKernelGGL(reset_draw_kernel, dim3((e->num_envs + 255) / 256), dim3(256), 0, st, *m, *e, *s, u_clip, reset_all);
  if (int rc = check_launch("reset_draw_kernel")) return rc;
  hipLaunchKernelGGL(env_reset_kernel, dim3(env_grid(e->num_envs)), dim3(64 * WAVES), 0, st, *m, *t, *e, *s, u_seg, u_jit, obs_out,
                     disc_obs_out, disc_demo_out, reset_all, head);
  return check_launch("env_reset_kernel");
}

extern "C" int addhip_motion_lookup(const addhip_motion_t* m, const int32_t* ids, const float* times, int32_t count,
                                    int32_t* idx_out, float* pose_out, float* vel_out, void* stream) {
  ADDHIP_REQUIRE(m && ids && times && count > 0, "bad lookup arguments");
  hipLaunchKernelGGL(lookup_kernel, dim3((count + 3) / 4), dim3(256), 0, (hipStream_t)stream, *m, ids, times, count, idx_out, pose_out, vel_out);
  return check_launch("lookup_kernel");
}

extern "C" int addhip_kin_engine_step(float* sim_pose, float* sim_vel, const float* target, int32_t target_stride, int32_t num_envs,
                                      float lag, float dt, void* stream) {
  ADDHIP_REQUIRE(sim_pose && sim_vel && target && num_envs > 0 && target_stride >= ADDHIP_NUM_DOF, "bad engine-step arguments");
  int n = num_envs * ADDHIP_NUM_DOF;
  hipLaunchKernelGGL(kin_engine_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sim_pose, sim_vel, target,
                     target_stride, num_envs, lag, dt);
  return check_launch("kin_engine_step_kernel");
}

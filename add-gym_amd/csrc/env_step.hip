// Per-env hot path of the rollout: reference-motion frame lookup, history ring, observation /
// discriminator-observation assembly, imitation reward, done flags, return tracker, masked reset.
//
// env_step_kernel: a wavefront owns a group of 1-16 consecutive envs and handles them one at a time.  The 16 rows an env
// needs (simulator pose+velocity, reference pose+velocity, up to 8 target frames, 2 older demo frames, 2 older history
// frames; 144 B each) are fetched in ONE pass -- 4 lanes per row, 36 B per lane -- into a wave-private LDS work area, the
// next env's rows already in flight; the few values that need arithmetic (tangent/normal vectors of 15 quaternions, target
// position offsets) are derived once per env by 15+24 lanes; every output row is then a pure gather from LDS through an
// index map built once per workgroup, stored 16 bytes per lane.  Reward / done / return tracker: the dof error sums of
// an env come from wave shuffles inside the loop, the scalar part runs after the loop with ONE LANE PER ENV.  HBM-bound by
// construction: 4.6 KB of algorithmic traffic per env-step.
// env_reset_kernel: masked reset, same staging/derive/emit code on clip rows.
//
// Reference functions restated here: see include/addhip.h at each entry point.
#include "common.h"
#include "record.h"
#include "quat.h"

using namespace addhip;

namespace {

constexpr int PW = ADDHIP_POSE_W;  // 36 floats per global row
constexpr int WAVES = 4;
// LDS rows: 16 staged in pass 0, 4 more (velocity history / demo velocity rows) in pass 1 when vel observations are on
constexpr int R_SIM = 0, R_SIMV = 1, R_REF = 2, R_REFV = 3, R_TAR = 4, R_DEMO0 = 12, R_DEMO1 = 13, R_H0 = 14, R_H1 = 15;
constexpr int R_D0V = 16, R_D1V = 17, R_H0V = 18, R_H1V = 19;
// four discriminator-observation steps (t.num_disc_obs_steps == 4): a third older demo / history row (+ their velocity rows), pass 1 too --
// such tasks run the two-pass instantiation whether or not they carry velocity observations
constexpr int R_DEMO2 = 20, R_H2 = 21, R_D2V = 22, R_H2V = 23;
constexpr int rows_of(bool vel) { return vel ? 24 : 16; }

// wave-private work area: derived values first, then the staged rows
constexpr int TN_CHAR = 0, TN_TAR = 1, TN_H0 = 9, TN_H1 = 10, TN_SIMG = 11, TN_D0 = 12, TN_D1 = 13, TN_REF = 14, TN_SLOTS = 15;
constexpr int TN_H2 = 15, TN_D2 = 16, TN_ALL = 17;  // (the two extra slots of a four-step task: derived by a second, small block)
constexpr int OFF_TN = 0;                          // [17][6] tangent/normal vectors
constexpr int OFF_TP = OFF_TN + TN_ALL * 6;        // [8][3] target position observations
// root velocity / angular velocity observations (heading-local when !global_obs) of 6 (rotation, velocity) row pairs
constexpr int TV_SIM = 0, TV_H0 = 1, TV_H1 = 2, TV_D0 = 3, TV_D1 = 4, TV_REF = 5, TV_SLOTS = 6;
constexpr int TV_H2 = 6, TV_D2 = 7, TV_ALL = 8;
constexpr int OFF_TV = OFF_TP + ADDHIP_MAX_TAR_STEPS * 3;   // [8][6]
constexpr int MAX_PHASE_ENC = 8;
constexpr int OFF_PH = OFF_TV + TV_ALL * 6;                 // phase, sin[P], cos[P]
constexpr int OFF_ZERO = OFF_PH + 1 + 2 * MAX_PHASE_ENC;
constexpr int OFF_ROWS = (OFF_ZERO + 1 + 3) / 4 * 4;        // 16-byte aligned
// LDS image of a row: 4 chunks of 9 floats padded to 12 so that each lane's chunk is 16-byte aligned
constexpr int LROW = 48;
__device__ __forceinline__ constexpr int row_off(int r, int c) { return OFF_ROWS + r * LROW + (c / 9) * 12 + (c % 9); }
constexpr int work_of(bool vel) { return OFF_ROWS + rows_of(vel) * LROW; }   // floats per wave
constexpr int map_of(bool vel, bool phase) { return vel ? 1024 : 640; }  // obs_stride + 2*disc_stride must fit

enum { K_SKIP = 0, K_POSE = 1, K_VEL = 2, K_SIM = 3, K_HIST = 4, K_SIMV = 5, K_HISTV = 6 };

// MotionLib.get_precomputed_motion_step index (anim/motion_lib.py:322-326): fp32 multiply by round(1/dt), truncate
// toward zero, clamp, add the clip offset.  Bit-exact: explicit round-to-nearest multiply, no contraction; the float is
// clamped to the int range first, which cannot change the clamped integer result.
__device__ __forceinline__ int step_index(float t, float dt_inv, int total_steps, int clip_start, int clip_steps, int compat) {
  float f = __fmul_rn(t, dt_inv);
  f = fminf(fmaxf(f, -1.0f), 2.0e9f);
  int fr = (int)f;  // truncation toward zero
  if (compat) {
    const int hi = total_steps - 1;
    fr = min(max(fr, 0), hi);
    return min(fr + clip_start, hi);  // (the reference would raise IndexError past the table end)
  }
  fr = min(max(fr, 0), clip_steps - 1);
  return fr + clip_start;
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

// ---- index maps: LDS source of every output element ------------------------------------------------------------
// compute_add_obs (add_observation.py:653-717 with compute_char_obs :422-459, compute_tar_obs :578-650)
__device__ __forceinline__ int obs_src(const addhip_task_t& t, int o) {
  if (o >= t.obs_dim) return OFF_ZERO;
  const int hc = t.root_height_obs ? 1 : 0;
  const int vw = t.enable_vel_obs ? 6 + ADDHIP_NUM_DOF : 0;
  const int char_dim = hc + 6 + ADDHIP_NUM_DOF + vw;
  if (o < char_dim) {
    if (hc && o == 0) return row_off(R_SIM, 2);
    int c = o - hc;
    if (c < 6) return OFF_TN + TN_CHAR * 6 + c;
    c -= 6;
    if (c < ADDHIP_NUM_DOF) return row_off(R_SIM, 7 + c);
    c -= ADDHIP_NUM_DOF;  // root_vel, root_ang_vel, dof_vel (add_observation.py:445-452)
    return c < 6 ? OFF_TV + TV_SIM * 6 + c : row_off(R_SIMV, 6 + c - 6);
  }
  const int ph_dim = t.enable_phase_obs ? 1 + 2 * t.num_phase_encoding : 0;  // add_observation.py:557-575
  if (o < char_dim + ph_dim) {
    const int i = o - char_dim;  // phase | sin terms | cos terms; OFF_PH holds them with stride MAX_PHASE_ENC
    if (i == 0) return OFF_PH;
    return i <= t.num_phase_encoding ? OFF_PH + i : OFF_PH + MAX_PHASE_ENC + (i - t.num_phase_encoding);
  }
  const int pw = hc ? 3 : 2;
  const int tw = pw + 6 + ADDHIP_NUM_DOF;
  const int k = (o - char_dim - ph_dim) / tw;
  int c = (o - char_dim - ph_dim) - k * tw;
  if (c < pw) return OFF_TP + k * 3 + c;
  c -= pw;
  return c < 6 ? OFF_TN + (TN_TAR + k) * 6 + c : row_off(R_TAR + k, 7 + c - 6);
}
// compute_disc_obs (add_observation.py:462-554); demo=false: history rows, true: clip rows
__device__ __forceinline__ int disc_src(const addhip_task_t& t, int o, bool demo) {
  if (o >= t.disc_dim) return OFF_ZERO;
  constexpr int pw = 3 + 6 + ADDHIP_NUM_DOF;  // 38
  const int sw = pw + (t.enable_vel_obs ? 6 + ADDHIP_NUM_DOF : 0);
  const int c = o - (o / sw) * sw;
  // step s of S = t.num_disc_obs_steps, oldest first: the last one (coded 3 here) is the current pose / the reference frame, the ones before
  // it the history rows / the earlier clip frames 0, 1, 2 (S = 2: one of each, S = 4: three)
  const int s = (o / sw == t.num_disc_obs_steps - 1) ? 3 : o / sw;
  const int row = demo ? (s == 0 ? R_DEMO0 : s == 1 ? R_DEMO1 : s == 2 ? R_DEMO2 : R_REF) : (s == 0 ? R_H0 : s == 1 ? R_H1 : s == 2 ? R_H2 : R_SIM);
  const int slot = demo ? (s == 0 ? TN_D0 : s == 1 ? TN_D1 : s == 2 ? TN_D2 : TN_REF) : (s == 0 ? TN_H0 : s == 1 ? TN_H1 : s == 2 ? TN_H2 : TN_SIMG);
  if (c < 3) return (!t.global_obs && c < 2) ? OFF_ZERO : row_off(row, c);
  if (c < 9) return OFF_TN + slot * 6 + (c - 3);
  if (c < pw) return row_off(row, 7 + c - 9);
  // compute_vel_obs (add_observation.py:502-517)
  const int cv = c - pw;
  const int vrow = demo ? (s == 0 ? R_D0V : s == 1 ? R_D1V : s == 2 ? R_D2V : R_REFV) : (s == 0 ? R_H0V : s == 1 ? R_H1V : s == 2 ? R_H2V : R_SIMV);
  const int vslot = demo ? (s == 0 ? TV_D0 : s == 1 ? TV_D1 : s == 2 ? TV_D2 : TV_REF) : (s == 0 ? TV_H0 : s == 1 ? TV_H1 : s == 2 ? TV_H2 : TV_SIM);
  return cv < 6 ? OFF_TV + vslot * 6 + cv : row_off(vrow, 6 + cv - 6);
}
__device__ __forceinline__ void build_maps(const addhip_task_t& t, short* maps) {
  const int n_obs = t.obs_stride, n_disc = t.disc_stride;
  for (int i = threadIdx.x; i < n_obs + 2 * n_disc; i += blockDim.x) {
    int v;
    if (i < n_obs) v = obs_src(t, i);
    else if (i < n_obs + n_disc) v = disc_src(t, i - n_obs, false);
    else v = disc_src(t, i - n_obs - n_disc, true);
    maps[i] = (short)v;
  }
}

// ---- per-lane staging role (fixed for the whole kernel): lane l copies floats [9q, 9q+9) of row r = l>>2, q = l&3 ----
// t.tar_dt[k] without dynamic indexing of the by-value kernel argument (which would be copied to scratch)
__device__ __forceinline__ float tar_dt_at(const addhip_task_t& t, int k) {
  float v = t.tar_dt[0];
#pragma unroll
  for (int i = 1; i < ADDHIP_MAX_TAR_STEPS; ++i) v = k == i ? t.tar_dt[i] : v;
  return v;
}

struct Tables {  // the motion-table fields the per-env code needs
  const float* pose; const float* vel; const int* clip_start; const int* clip_steps;
  int total_steps, compat; float dt_inv;
};

struct EnvPtrs { const float* sim_pose; const float* sim_vel; const float* hist; const float* hist_vel; };

// What a lane copies is fixed for the whole kernel, so the source is resolved once: a base pointer (already advanced to
// the lane's 9-float chunk and, for history rows, to its ring slot) plus a per-env row multiplier.
struct Role {
  int kind;           // ROLE_SKIP, ROLE_TABLE (row = motion-table step index) or ROLE_ENV (row = env)
  float dt;           // time offset of a table row relative to the env's motion time
  const float* base;
  int mul;            // floats per row index
};
enum { ROLE_SKIP = 0, ROLE_TABLE = 1, ROLE_ENV = 2 };

// FRESH (reset): every row comes from the clip tables, the env pointers are never touched
template <bool VEL, bool FRESH>
__device__ __forceinline__ Role lane_role(const addhip_task_t& t, const Tables& tb, const EnvPtrs& ep, int lane, int pass, int h0, int h1, int h2) {
  constexpr bool fresh = FRESH;
  const bool three = t.num_disc_obs_steps >= 3, four = t.num_disc_obs_steps == 4;  // (two steps: the second history / clip row is not staged)
  const bool vel_on = VEL && t.enable_vel_obs;  // (a four-step task without velocity observations runs the two-pass instantiation too)
  const int r = pass * 16 + (lane >> 2), q = lane & 3;
  int kind = K_SKIP, hslot = 0;
  float dt = 0.0f;
  if (r == R_SIM) kind = fresh ? K_POSE : K_SIM;
  else if (r == R_REF) kind = K_POSE;
  else if (r == R_REFV) kind = K_VEL;
  else if (r >= R_TAR && r < R_TAR + ADDHIP_MAX_TAR_STEPS) {
    const int k = r - R_TAR;
    if (k < t.num_tar_steps) { kind = K_POSE; dt = tar_dt_at(t, k); }  // add_observation.py:214-215
  } else if (r == R_DEMO0 || (r == R_H0 && fresh)) { kind = K_POSE; dt = t.demo_dt[0]; }   // :362-375
  else if (r == R_DEMO1 || (r == R_H1 && fresh)) { if (three) { kind = K_POSE; dt = t.demo_dt[1]; } }
  else if (r == R_H0) { kind = K_HIST; hslot = h0; }
  else if (r == R_H1) { if (three) { kind = K_HIST; hslot = h1; } }
  else if (r == R_SIMV) { if (vel_on || !FRESH) kind = fresh ? K_VEL : K_SIMV; }  // the step's reward reads it too
  else if (VEL) {
    if (r == R_DEMO2 || (r == R_H2 && fresh)) { if (four) { kind = K_POSE; dt = t.demo_dt[2]; } }
    else if (r == R_H2) { if (four) { kind = K_HIST; hslot = h2; } }
    else if (vel_on) {
      if (r == R_D0V || (r == R_H0V && fresh)) { kind = K_VEL; dt = t.demo_dt[0]; }
      else if (r == R_D1V || (r == R_H1V && fresh)) { if (three) { kind = K_VEL; dt = t.demo_dt[1]; } }
      else if (r == R_D2V || (r == R_H2V && fresh)) { if (four) { kind = K_VEL; dt = t.demo_dt[2]; } }
      else if (r == R_H0V) { kind = K_HISTV; hslot = h0; }
      else if (r == R_H1V) { if (three) { kind = K_HISTV; hslot = h1; } }
      else if (r == R_H2V) { if (four) { kind = K_HISTV; hslot = h2; } }
    }
  }
  Role ro{ROLE_SKIP, dt, nullptr, PW};
  if (kind == K_SKIP) return ro;
  // base pointer by masks, not by a select chain (which the optimiser turns into a pointer table in scratch memory)
  auto pick = [](bool c, const float* p) -> unsigned long long { return c ? (unsigned long long)p : 0ull; };
  unsigned long long b = pick(kind == K_POSE, tb.pose) | pick(kind == K_VEL, tb.vel);
  if (!FRESH) b |= pick(kind == K_SIM, ep.sim_pose) | pick(kind == K_SIMV, ep.sim_vel) | pick(kind == K_HIST, ep.hist) | pick(kind == K_HISTV, ep.hist_vel);
  const bool table = FRESH || kind == K_POSE || kind == K_VEL, ring = kind == K_HIST || kind == K_HISTV;
  ro.kind = table ? ROLE_TABLE : ROLE_ENV;
  ro.base = reinterpret_cast<const float*>(b) + (ring ? hslot * PW : 0) + q * 9;
  ro.mul = ring ? t.num_disc_obs_steps * PW : PW;
  return ro;
}

__device__ __forceinline__ void stage_rows(float* w, const Role& ro, const Tables& tb, int pass, int env, int cstart, int csteps,
                                           float tm, int lane) {
  if (ro.kind == ROLE_SKIP) return;
  int row = env;
  if (ro.kind == ROLE_TABLE) row = step_index(__fadd_rn(tm, ro.dt), tb.dt_inv, tb.total_steps, cstart, csteps, tb.compat);
  const float* src = ro.base + (size_t)(unsigned)(row * ro.mul);  // < 2^31 floats (check_common)
  // nine dword loads (this granularity is only 4-byte aligned); the backend merges them into 16+16+4 byte loads
  float v[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = src[i];
  float* dst = w + OFF_ROWS + (pass * 16 + (lane >> 2)) * LROW + (lane & 3) * 12;
  *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
  dst[8] = v[8];
}

// the two halves of stage_rows, so that the loads of the next env can be in flight while the current one is emitted
struct RowRegs { float v[9]; };
__device__ __forceinline__ void load_rows(RowRegs& rr, const Role& ro, const Tables& tb, int env, int cstart, int csteps, float tm) {
  if (ro.kind == ROLE_SKIP) return;
  int row = env;
  if (ro.kind == ROLE_TABLE) row = step_index(__fadd_rn(tm, ro.dt), tb.dt_inv, tb.total_steps, cstart, csteps, tb.compat);
  const float* src = ro.base + (size_t)(unsigned)(row * ro.mul);
#pragma unroll
  for (int i = 0; i < 9; ++i) rr.v[i] = src[i];
}
__device__ __forceinline__ void store_rows(float* w, const RowRegs& rr, const Role& ro, int pass, int lane) {
  if (ro.kind == ROLE_SKIP) return;
  float* dst = w + OFF_ROWS + (pass * 16 + (lane >> 2)) * LROW + (lane & 3) * 12;
  *reinterpret_cast<float4*>(dst) = make_float4(rr.v[0], rr.v[1], rr.v[2], rr.v[3]);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(rr.v[4], rr.v[5], rr.v[6], rr.v[7]);
  dst[8] = rr.v[8];
}

__device__ __forceinline__ Quat lds_quat(const float* w, int r) {
  return Quat{w[row_off(r, 3)], w[row_off(r, 4)], w[row_off(r, 5)], w[row_off(r, 6)]};
}

// velocity / angular velocity of row rv in the heading frame of row rr (or the global frame) -> slot s of OFF_TV
template <bool GLOBAL>
__device__ __forceinline__ void vel_slot(float* w, int s, int rr, int rv) {
  Vec3 v{w[row_off(rv, 0)], w[row_off(rv, 1)], w[row_off(rv, 2)]}, a{w[row_off(rv, 3)], w[row_off(rv, 4)], w[row_off(rv, 5)]};
  if (!GLOBAL) {
    const Quat h = heading_quat_inv(lds_quat(w, rr));
    v = quat_rotate(h, v);
    a = quat_rotate(h, a);
  }
  float* d = w + OFF_TV + s * 6;
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = a.x; d[4] = a.y; d[5] = a.z;
}

// per-env arithmetic, once: lane s < 15 -> tangent+normal (torch_util.py:231-242) of quaternion slot s;
// lanes 16.. -> target position observations (add_observation.py:589-618)
template <bool GLOBAL, bool VEL, bool PHASE>
__device__ __forceinline__ void derive(const addhip_task_t& t, float* w, int lane, float phase) {
  if (lane < TN_SLOTS) {
    const int slot = lane;
    int r;
    bool local = false;
    if (slot == TN_CHAR) { r = R_SIM; local = !GLOBAL; }
    else if (slot < TN_TAR + ADDHIP_MAX_TAR_STEPS) { r = R_TAR + (slot - TN_TAR); local = !GLOBAL; }
    else r = slot == TN_H0 ? R_H0 : slot == TN_H1 ? R_H1 : slot == TN_SIMG ? R_SIM : slot == TN_D0 ? R_DEMO0 : slot == TN_D1 ? R_DEMO1 : R_REF;
    Quat q = lds_quat(w, r);
    if (local) q = quat_mul(heading_quat_inv(lds_quat(w, slot == TN_CHAR ? R_SIM : R_TAR)), q);
    const Vec3 tan = quat_rotate(q, Vec3{1.0f, 0.0f, 0.0f}), nrm = quat_rotate(q, Vec3{0.0f, 0.0f, 1.0f});
    float* d = w + OFF_TN + slot * 6;
    d[0] = tan.x; d[1] = tan.y; d[2] = tan.z; d[3] = nrm.x; d[4] = nrm.y; d[5] = nrm.z;
  } else if (lane >= 16 && lane < 16 + t.num_tar_steps * 3) {
    const int i = lane - 16, k = i / 3, c = i - k * 3;
    const int r = R_TAR + k;
    float v;
    if (c == 2) v = w[row_off(r, 2)];  // add_observation.py:615-616 absolute height
    else if (GLOBAL) v = w[row_off(r, c)] - w[row_off(R_SIM, c)];
    else {
      Vec3 d{w[row_off(r, 0)] - w[row_off(R_TAR, 0)], w[row_off(r, 1)] - w[row_off(R_TAR, 1)], w[row_off(r, 2)] - w[row_off(R_TAR, 2)]};
      const Vec3 rr = quat_rotate(heading_quat_inv(lds_quat(w, R_TAR)), d);
      v = c == 0 ? rr.x : rr.y;
    }
    w[OFF_TP + i] = v;
  } else if (lane >= 40 && lane < 40 + TV_SLOTS) {
    if (VEL && t.enable_vel_obs) {  // compute_char_obs / compute_vel_obs velocity terms (add_observation.py:445-452, 502-517)
      const int s = lane - 40;
      const int rr = s == TV_SIM ? R_SIM : s == TV_H0 ? R_H0 : s == TV_H1 ? R_H1 : s == TV_D0 ? R_DEMO0 : s == TV_D1 ? R_DEMO1 : R_REF;
      const int rv = s == TV_SIM ? R_SIMV : s == TV_H0 ? R_H0V : s == TV_H1 ? R_H1V : s == TV_D0 ? R_D0V : s == TV_D1 ? R_D1V : R_REFV;
      vel_slot<GLOBAL>(w, s, rr, rv);
    }
  } else if (lane >= 46 && lane < 47 + 2 * MAX_PHASE_ENC) {
    if (PHASE) {  // compute_phase_obs (add_observation.py:557-575)
      const int i = lane - 46;
      if (i == 0) w[OFF_PH] = phase;
      else {
        const int e = (i - 1) % MAX_PHASE_ENC;
        if (e < t.num_phase_encoding) {
          const float val = __fmul_rn(phase, __fmul_rn(6.2831855f, (float)(1 << e)));  // 2*pi*2^e in fp32
          w[OFF_PH + i] = i <= MAX_PHASE_ENC ? sinf(val) : cosf(val);
        }
      }
    }
  } else if (lane == 63) {
    w[OFF_ZERO] = 0.0f;
  }
  if (VEL && t.num_disc_obs_steps == 4) {  // the third older history / clip row of a four-step task: two more quaternion slots, two more velocity slots
    if (lane < 2) {
      const Quat q = lds_quat(w, lane == 0 ? R_H2 : R_DEMO2);
      const Vec3 tan = quat_rotate(q, Vec3{1.0f, 0.0f, 0.0f}), nrm = quat_rotate(q, Vec3{0.0f, 0.0f, 1.0f});
      float* d = w + OFF_TN + (lane == 0 ? TN_H2 : TN_D2) * 6;
      d[0] = tan.x; d[1] = tan.y; d[2] = tan.z; d[3] = nrm.x; d[4] = nrm.y; d[5] = nrm.z;
    } else if (lane < 4 && t.enable_vel_obs) {
      vel_slot<GLOBAL>(w, lane == 2 ? TV_H2 : TV_D2, lane == 2 ? R_H2 : R_DEMO2, lane == 2 ? R_H2V : R_D2V);
    }
  }
}

// n is a multiple of 4 and rows are 16-byte aligned: one 16-byte store per lane per pass
__device__ __forceinline__ void emit(const float* w, const short* map, int n, float* out, int lane) {
  for (int o = lane * 4; o < n; o += 256) {
    const short4 m = *reinterpret_cast<const short4*>(map + o);
    *reinterpret_cast<float4*>(out + o) = make_float4(w[m.x], w[m.y], w[m.z], w[m.w]);
  }
}

// wave-private LDS hand-off: DS operations of one wave execute in program order, so only the compiler must be kept
// from moving accesses across the phase boundary
__device__ __forceinline__ void wave_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

struct StepArgs {
  Tables tb;
  const float* sim_pose; const float* sim_vel; float* time; const float* time_off; const int* motion_id; float* hist; float* hist_vel;
  const float* clip_len; const int* clip_loop;
  int* done; const unsigned char* contact; float* ret_acc; int* len_acc; const float* dof_err_w;
  float* ref_pose; float* ref_vel;
  float* obs; float* obs2; float* obs_timeout; float* disc; float* demo;
  float* reward; int* done_rec; int* motion_id_rec; float* motion_time_rec; float* ep_stats;
  int num_envs, head, envs_per_wave;
};

// ---- reward (add_reward.py:103-177), done (add_done.py:96-147), return tracker (base_agent.py:596-621) ---------
// Per env the wave leaves 31 values in an LDS slot: the 26 root values of simulator and reference (sim pos3 quat4 |
// ref pos3 quat4 | sim vel3 ang3 | ref vel3 ang3), the two dof error sums, the clocks and the clip id.  Once per
// group of up to SLOT_MAX consecutive envs, ONE LANE PER ENV does the scalar part (quaternion angle, four exp, flags,
// tracker): its ~300 instructions are paid once per group instead of once per env, and its loads / stores of the
// per-env scalars are coalesced.
constexpr int SLOT_MAX = 16, SLOT_W = 33;  // odd stride: conflict-free lane-per-slot reads
enum { S_PE = 26, S_VE = 27, S_TIME = 28, S_TM = 29, S_ID = 30, S_PEU = 31 };  // PE/VE: weighted sums (reward); PEU: unweighted pose sum (done)

// LDS source of root value j (0..25) in the staged rows
__device__ __forceinline__ int root_src(int j) {
  if (j < 7) return row_off(R_SIM, j);
  if (j < 14) return row_off(R_REF, j - 7);
  if (j < 20) return row_off(R_SIMV, j - 14);
  return row_off(R_REFV, j - 20);
}

// returns the wave-wide mask of lanes (= envs of the group) that finished with DONE_TIME
template <bool GLOBAL>
__device__ __forceinline__ unsigned long long reward_done_group(const addhip_task_t& t, const StepArgs& a, const float* slots, int first_env,
                                                                int count, int lane) {
  const bool valid = lane < count;
  const int env = first_env + (valid ? lane : 0);
  const float* rt = slots + (valid ? lane : 0) * SLOT_W;
  const bool contact = a.contact ? (a.contact[env] != 0) : false;
  const float ret_old = a.ret_acc ? a.ret_acc[env] : 0.0f;
  const int len_old = a.ret_acc ? a.len_acc[env] : 0;
  const int id = __float_as_int(rt[S_ID]);
  const float clip_len = a.clip_len[id];
  const int clip_loop = a.clip_loop[id];
  const float pe = rt[S_PE], ve = rt[S_VE], pe_unweighted = rt[S_PEU], time_new = rt[S_TIME], tm = rt[S_TM];
  const float dx = rt[7] - rt[0], dy = rt[8] - rt[1], dz = rt[9] - rt[2];
  Quat q_sim{rt[3], rt[4], rt[5], rt[6]}, q_ref{rt[10], rt[11], rt[12], rt[13]};
  Vec3 v_sim{rt[14], rt[15], rt[16]}, w_sim{rt[17], rt[18], rt[19]}, v_ref{rt[20], rt[21], rt[22]}, w_ref{rt[23], rt[24], rt[25]};
  const bool track_root = (t.num_tar_steps > 0) && GLOBAL;  // add_observation.py:349-350
  const float root_err_full = dx * dx + dy * dy + dz * dz;
  const float rx = track_root ? dx : 0.0f, ry = track_root ? dy : 0.0f, rz = t.root_height_obs ? dz : 0.0f;
  const float root_pos_err = rx * rx + ry * ry + rz * rz;
  if (!track_root) {  // convert_to_local_root (add_reward.py:91-101)
    const Quat hs = heading_quat_inv(q_sim), hr = heading_quat_inv(q_ref);
    v_sim = quat_rotate(hs, v_sim); w_sim = quat_rotate(hs, w_sim); q_sim = quat_mul(hs, q_sim);
    v_ref = quat_rotate(hr, v_ref); w_ref = quat_rotate(hr, w_ref); q_ref = quat_mul(hr, q_ref);
  }
  float rot_err = quat_diff_angle(q_sim, q_ref);
  rot_err *= rot_err;
  const float vx = v_ref.x - v_sim.x, vy = v_ref.y - v_sim.y, vz = v_ref.z - v_sim.z;
  const float root_vel_err = vx * vx + vy * vy + vz * vz;
  const float ax = w_ref.x - w_sim.x, ay = w_ref.y - w_sim.y, az = w_ref.z - w_sim.z;
  const float root_ang_err = ax * ax + ay * ay + az * az;
  const float r = t.pose_w * expf(-t.pose_scale * pe) + t.vel_w * expf(-t.vel_scale * ve) +
                  t.root_pose_w * expf(-t.root_pose_scale * (root_pos_err + 0.1f * rot_err)) +
                  t.root_vel_w * expf(-t.root_vel_scale * (root_vel_err + 0.1f * root_ang_err));
  int done = ADDHIP_DONE_NULL;
  if (time_new >= t.max_episode_length) done = ADDHIP_DONE_TIME;
  if (tm >= clip_len && clip_loop != 1) done = ADDHIP_DONE_SUCC;
  if (t.enable_early_termination) {
    bool failed = contact;
    if (t.pose_termination) {
      bool pose_fail = (pe_unweighted / (float)ADDHIP_NUM_DOF) > t.pose_termination_dist;  // add_done.py:129-132: plain mean
      if (track_root) pose_fail = pose_fail || (root_err_full > t.pose_termination_dist);
      failed = failed || pose_fail;
    }
    if (failed && time_new > 0.0f) done = ADDHIP_DONE_FAIL;
  }
  if (valid) {
    a.time[env] = time_new;
    a.done[env] = done;
    if (a.done_rec) a.done_rec[env] = done;
    if (a.reward) a.reward[env] = r;
    if (a.motion_id_rec) a.motion_id_rec[env] = id;
    if (a.motion_time_rec) a.motion_time_rec[env] = tm;
  }
  if (a.ret_acc) {
    const float ra = ret_old + r;
    const int la = len_old + 1;
    const bool finished = valid && done != ADDHIP_DONE_NULL;
    if (a.ep_stats && __ballot(finished)) {  // one atomic triple per group, not per finished episode
      const float s_ret = wave_sum(finished ? ra : 0.0f), s_len = wave_sum(finished ? (float)la : 0.0f), s_cnt = wave_sum(finished ? 1.0f : 0.0f);
      if (lane == 0) {
        atomicAdd(&a.ep_stats[0], s_ret);
        atomicAdd(&a.ep_stats[1], s_len);
        atomicAdd(&a.ep_stats[2], s_cnt);
      }
    }
    if (valid) {
      a.ret_acc[env] = finished ? 0.0f : ra;
      a.len_acc[env] = finished ? 0 : la;
    }
  }
  return __ballot(valid && done == ADDHIP_DONE_TIME);
}

template <bool GLOBAL, bool VEL, bool PHASE>
__global__ __launch_bounds__(64 * WAVES) void env_step_kernel(addhip_task_t t, StepArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[WAVES][work_of(VEL)];
  __shared__ float slot_mem[WAVES][SLOT_MAX * SLOT_W];
  __shared__ short maps[map_of(VEL, PHASE)];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* w = lds[wv];
  float* slots = slot_mem[wv];
  build_maps(t, maps);
  const int S = t.num_disc_obs_steps;  // ring depth: slot a.head receives this step's pose, the others hold the S - 1 poses before it, oldest first from head + 1
  const int h0 = (a.head + 1) % S, h1 = (a.head + 2) % S, h2 = (a.head + 3) % S;
  const EnvPtrs ep{a.sim_pose, a.sim_vel, a.hist, a.hist_vel};
  const Role ro = lane_role<VEL, false>(t, a.tb, ep, lane, 0, h0, h1, h2);
  Role ro1{ROLE_SKIP, 0.0f, nullptr, PW};
  if (VEL) ro1 = lane_role<VEL, false>(t, a.tb, ep, lane, 1, h0, h1, h2);
  const int n_obs = t.obs_stride, n_disc = t.disc_stride;
  // loop-invariant LDS offsets of this lane's reward inputs: lanes 0-28 dof positions, 32-60 dof velocities; lanes
  // 0-25 also copy one root value each
  const int hl = lane & 31;
  const bool dof_lane = hl < ADDHIP_NUM_DOF;
  const int dof_a = lane < 32 ? row_off(R_REF, 7 + (dof_lane ? hl : 0)) : row_off(R_REFV, 6 + (dof_lane ? hl : 0));
  const int dof_b = lane < 32 ? row_off(R_SIM, 7 + (dof_lane ? hl : 0)) : row_off(R_SIMV, 6 + (dof_lane ? hl : 0));
  const int root_off = root_src(lane < 26 ? lane : 0);
  const float dof_w = (a.dof_err_w && dof_lane) ? a.dof_err_w[hl] : 1.0f;  // add_reward.py:28-52 (joint weights, per dof)
  __syncthreads();  // maps; from here on the waves of a workgroup run independently
  // a wave owns ONE group of envs_per_wave consecutive envs (no outer loop: the scalar phase after the env loop then
  // starts from an empty register file instead of having its constants hoisted across the loop)
  {
    const int first = (blockIdx.x * WAVES + wv) * a.envs_per_wave;
    const int count = min(a.envs_per_wave, a.num_envs - first);
    if (count <= 0) return;
    // per-env scalars of the whole group, lane k <-> env first+k: two dependent loads per GROUP instead of per env
    const int mine = first + min(lane, count - 1);
    const float g_time = __fadd_rn(a.time[mine], t.dt);       // env.py:155
    const int g_id = a.motion_id[mine];
    const float g_tm = __fadd_rn(g_time, a.time_off[mine]);   // add_observation.py:352-354
    const int g_cstart = a.tb.clip_start[g_id], g_csteps = a.tb.clip_steps[g_id];
    float g_phase = 0.0f;
    if (PHASE) g_phase = fminf(fmaxf(__fdiv_rn(g_tm, a.clip_len[g_id]), 0.0f), 1.0f);  // motion_lib.py:361-372 (CLAMP clips)
    RowRegs r0, r1;
    load_rows(r0, ro, a.tb, first, __builtin_amdgcn_readlane(g_cstart, 0), __builtin_amdgcn_readlane(g_csteps, 0),
              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_tm), 0)));
    if (VEL) load_rows(r1, ro1, a.tb, first, __builtin_amdgcn_readlane(g_cstart, 0), __builtin_amdgcn_readlane(g_csteps, 0),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_tm), 0)));
    for (int k = 0; k < count; ++k) {
      const int env = first + k;
      const float time_new = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_time), k));
      const float tm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_tm), k));
      const int id = __builtin_amdgcn_readlane(g_id, k);
      const float phase = PHASE ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_phase), k)) : 0.0f;
      store_rows(w, r0, ro, 0, lane);
      if (VEL) store_rows(w, r1, ro1, 1, lane);
      wave_sync();
      if (k + 1 < count) {  // next env's rows: in flight during derive + emit of this one
        const int cs = __builtin_amdgcn_readlane(g_cstart, k + 1), cn = __builtin_amdgcn_readlane(g_csteps, k + 1);
        const float tn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_tm), k + 1));
        load_rows(r0, ro, a.tb, env + 1, cs, cn, tn);
        if (VEL) load_rows(r1, ro1, a.tb, env + 1, cs, cn, tn);
      }
      derive<GLOBAL, VEL, PHASE>(t, w, lane, phase);
      {  // reward inputs of this env -> slot k
        float* sl = slots + k * SLOT_W;
        float squ = 0.0f;
        if (dof_lane) { const float d = w[dof_a] - w[dof_b]; squ = d * d; }
        float sq = squ * dof_w;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { sq += __shfl_xor(sq, o, 64); squ += __shfl_xor(squ, o, 64); }
        if (lane < 26) sl[lane] = w[root_off];
        if (lane == 0) { sl[S_PE] = sq; sl[S_PEU] = squ; sl[S_TIME] = time_new; sl[S_TM] = tm; sl[S_ID] = __int_as_float(id); }
        if (lane == 32) sl[S_VE] = sq;
      }
      wave_sync();
      if (a.obs) emit(w, maps, n_obs, a.obs + (size_t)env * n_obs, lane);
      if (a.obs2) emit(w, maps, n_obs, a.obs2 + (size_t)env * n_obs, lane);
      if (a.disc) emit(w, maps + n_obs, n_disc, a.disc + (size_t)env * n_disc, lane);
      if (a.demo) emit(w, maps + n_obs + n_disc, n_disc, a.demo + (size_t)env * n_disc, lane);
      if (lane < PW) {
        // history push (circular_buffer.py:17-20) and, optionally, the reference state (add_observation.py:163-174)
        a.hist[((size_t)env * S + a.head) * PW + lane] = w[row_off(R_SIM, lane)];
        if (VEL && t.enable_vel_obs) a.hist_vel[((size_t)env * S + a.head) * PW + lane] = w[row_off(R_SIMV, lane)];
        if (a.ref_pose) a.ref_pose[(size_t)env * PW + lane] = w[row_off(R_REF, lane)];
        if (a.ref_vel) a.ref_vel[(size_t)env * PW + lane] = w[row_off(R_REFV, lane)];
      }
      wave_sync();
    }
    unsigned long long timed_out = reward_done_group<GLOBAL>(t, a, slots, first, count, lane);
    if (timed_out && a.obs_timeout && a.obs) {
      // rare (once per max_episode_length per env): keep the pre-reset obs row of these envs for the critic's
      // next-value (ppo_agent.py:117-133); the reset kernel is about to overwrite it in the obs slot.  The rows were
      // stored by this wave above: make them visible to its own loads first.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      while (timed_out) {
        const int k = __ffsll((long long)timed_out) - 1;
        timed_out &= timed_out - 1;
        const float* src = a.obs + (size_t)(first + k) * n_obs;
        float* dst = a.obs_timeout + (size_t)(first + k) * n_obs;
        for (int o = lane; o < n_obs; o += 64) dst[o] = __builtin_nontemporal_load(src + o);
      }
    }
  }
}

// ---- reset phase 1: clip draw (MotionLib.sample_motions, motion_lib.py:35-39) + batch temperature
__global__ void reset_draw_kernel(int num_envs, int num_clips, const int* done, int* motion_id, addhip_sampler_t s, const float* u_clip,
                                  int reset_all) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= num_envs) return;
  if (!reset_all && done[env] == ADDHIP_DONE_NULL) return;
  const float u = u_clip[env];
  int id = 0;
  while (id < num_clips - 1 && !(u < s.clip_cdf[id])) ++id;
  motion_id[env] = id;
  if (s.temperature <= 0.0f) {
    float mx = 0.0f;  // errors are >= 0
    for (int k = 0; k < s.num_segments; ++k) mx = fmaxf(mx, s.errors[id * s.num_segments + k]);
    atomicMax(s.temp_bits, __float_as_uint(mx));
  }
}

// c10::div_floor_floating, the arithmetic behind `time // dt` (sampler.py:88)
__device__ __forceinline__ float floor_div_f32(float a, float b) {
  const float mod = fmodf(a, b);
  float div = __fdiv_rn(__fsub_rn(a, mod), b);
  if (mod != 0.0f && ((b < 0.0f) != (mod < 0.0f))) div = __fsub_rn(div, 1.0f);
  if (div != 0.0f) {
    float fl = floorf(div);
    if (__fsub_rn(div, fl) > 0.5f) fl = __fadd_rn(fl, 1.0f);
    return fl;
  }
  return copysignf(0.0f, __fdiv_rn(a, b));
}

struct ResetArgs {
  Tables tb;
  const float* clip_len;
  float* sim_pose; float* sim_vel; float* time; float* time_off; const int* motion_id; float* hist; float* hist_vel; int* done;
  float* ref_pose; float* ref_vel;
  const float* u_seg; const float* u_jit;
  float* obs; float* disc; float* demo;
  int num_envs, head, reset_all;
};

// ---- reset phase 2: start time, state from the clip, history refill, observations
template <bool GLOBAL, bool VEL, bool PHASE>
__global__ __launch_bounds__(64 * WAVES) void env_reset_kernel(addhip_task_t t, addhip_sampler_t s, ResetArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[WAVES][work_of(VEL)];
  __shared__ short maps[map_of(VEL, PHASE)];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* w = lds[wv];
  build_maps(t, maps);
  const EnvPtrs ep{nullptr, nullptr, nullptr, nullptr};
  const Role ro = lane_role<VEL, true>(t, a.tb, ep, lane, 0, 0, 0, 0);
  Role ro1{ROLE_SKIP, 0.0f, nullptr, PW};
  if (VEL) ro1 = lane_role<VEL, true>(t, a.tb, ep, lane, 1, 0, 0, 0);
  const int n_obs = t.obs_stride, n_disc = t.disc_stride;
  __syncthreads();  // maps
  for (int env = blockIdx.x * WAVES + wv; env < a.num_envs; env += gridDim.x * WAVES) {
    if (!a.reset_all && a.done[env] == ADDHIP_DONE_NULL) continue;  // wave-uniform
    float off = 0.0f, phase = 0.0f;
    {
      const int id = a.motion_id[env];
      if (s.rand_reset) {
        // AdaptiveSegmentSampler.get_probs + multinomial by inverse CDF (sampler.py:57-80)
        const int S = s.num_segments;
        const float temp = s.temperature > 0.0f ? s.temperature : __fadd_rn(__uint_as_float(*s.temp_bits), 1e-6f);
        const float z = lane < S ? s.errors[id * S + lane] / temp : -INFINITY;
        const float zmax = wave_max(z);
        const float ez = lane < S ? expf(z - zmax) : 0.0f;
        const float p = ez / wave_sum(ez);
        float cdf = p;  // inclusive scan over the first S lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const float up = __shfl_up(cdf, d, 64);
          if (lane >= d) cdf += up;
        }
        const float u = a.u_seg[env];
        int seg = __popcll(__ballot(lane < S && !(u < cdf)));
        if (seg > S - 1) seg = S - 1;
        const float ss = s.seg_size[id];
        float tt = __fmul_rn((float)seg, ss);                   // sampler.py:81-82
        tt = __fadd_rn(tt, __fmul_rn(a.u_jit[env], ss));        // :84-85
        tt = __fmul_rn(floor_div_f32(tt, t.dt), t.dt);          // :88
        off = fmaxf(tt, s.min_start_time);                      // :91
      }
      // time_buf = 0 (env.py:161) -> motion time == offset
      const int cstart = a.tb.clip_start[id], csteps = a.tb.clip_steps[id];
      stage_rows(w, ro, a.tb, 0, env, cstart, csteps, off, lane);
      if (VEL) stage_rows(w, ro1, a.tb, 1, env, cstart, csteps, off, lane);
      if (PHASE) phase = fminf(fmaxf(__fdiv_rn(off, a.clip_len[id]), 0.0f), 1.0f);
    }
    wave_sync();
    derive<GLOBAL, VEL, PHASE>(t, w, lane, phase);
    wave_sync();
    {
      if (a.obs) emit(w, maps, n_obs, a.obs + (size_t)env * n_obs, lane);
      if (a.disc) emit(w, maps + n_obs, n_disc, a.disc + (size_t)env * n_disc, lane);
      if (a.demo) emit(w, maps + n_obs + n_disc, n_disc, a.demo + (size_t)env * n_disc, lane);
      if (lane < PW) {
        const float pose = w[row_off(R_REF, lane)], vel = w[row_off(R_REFV, lane)];
        // set_qpos / set_dofs_velocity payload (add_observation.py:314-331) straight into the simulator state
        a.sim_pose[(size_t)env * PW + lane] = pose;
        a.sim_vel[(size_t)env * PW + lane] = vel;
        if (a.ref_pose) a.ref_pose[(size_t)env * PW + lane] = pose;
        if (a.ref_vel) a.ref_vel[(size_t)env * PW + lane] = vel;
        // CircularBuffer.fill (circular_buffer.py:22-29): get_all() order = demo frames t-2dt, t-dt, t
        // (S = t.num_disc_obs_steps slots: S = 2 holds t-dt, t; S = 4: t-3dt .. t)
        const int S = t.num_disc_obs_steps;
        float* hb = a.hist + (size_t)env * S * PW;
        hb[((a.head + 0) % S) * PW + lane] = w[row_off(R_DEMO0, lane)];
        if (S >= 3) hb[((a.head + 1) % S) * PW + lane] = w[row_off(R_DEMO1, lane)];
        if (VEL && S == 4) hb[((a.head + 2) % S) * PW + lane] = w[row_off(R_DEMO2, lane)];
        hb[((a.head + S - 1) % S) * PW + lane] = pose;
        if (VEL && t.enable_vel_obs) {
          float* hv = a.hist_vel + (size_t)env * S * PW;
          hv[((a.head + 0) % S) * PW + lane] = w[row_off(R_D0V, lane)];
          if (S >= 3) hv[((a.head + 1) % S) * PW + lane] = w[row_off(R_D1V, lane)];
          if (S == 4) hv[((a.head + 2) % S) * PW + lane] = w[row_off(R_D2V, lane)];
          hv[((a.head + S - 1) % S) * PW + lane] = vel;
        }
      }
      if (lane == 0) {
        a.time[env] = 0.0f;
        a.time_off[env] = off;
        a.done[env] = ADDHIP_DONE_NULL;  // add_done.py:92-93
      }
    }
    wave_sync();
  }
}

__global__ void lookup_kernel(addhip_motion_t m, const int* ids, const float* times, int count, int* idx_out, float* pose_out, float* vel_out) {
  const int q = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= count) return;
  const int id = ids[q];
  const int idx = step_index(times[q], m.dt_inv, m.total_steps, m.clip_start[id], m.clip_steps[id], m.reference_compat);
  if (lane == 0 && idx_out) idx_out[q] = idx;
  if (lane < PW) {
    if (pose_out) pose_out[(size_t)q * PW + lane] = m.pose[(size_t)idx * PW + lane];
    if (vel_out) vel_out[(size_t)q * PW + lane] = m.vel[(size_t)idx * PW + lane];
  }
}

__global__ void kin_engine_step_kernel(float* sim_pose, float* sim_vel, const float* target, int tstride, int n, float lag, float dt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ADDHIP_NUM_DOF) return;
  const int env = i / ADDHIP_NUM_DOF, j = i - env * ADDHIP_NUM_DOF;
  const float q = sim_pose[(size_t)env * PW + 7 + j];
  const float qn = __fadd_rn(q, __fmul_rn(lag, __fsub_rn(target[(size_t)env * tstride + j], q)));
  sim_vel[(size_t)env * PW + 6 + j] = __fdiv_rn(__fsub_rn(qn, q), dt);
  sim_pose[(size_t)env * PW + 7 + j] = qn;
}

int check_common(const addhip_motion_t* m, const addhip_task_t* t, const addhip_env_t* e) {
  ADDHIP_REQUIRE(m && t && e, "null struct");
  ADDHIP_REQUIRE(e->num_envs > 0, "num_envs must be > 0");
  ADDHIP_REQUIRE(m->pose && m->vel && m->clip_start && m->clip_len && m->clip_loop && m->clip_steps, "motion tables missing");
  ADDHIP_REQUIRE(m->total_steps > 0 && m->num_clips > 0, "empty motion library");
  ADDHIP_REQUIRE((long long)m->total_steps * PW < (1ll << 31), "motion table too large for 32-bit row offsets");
  ADDHIP_REQUIRE((long long)e->num_envs * ADDHIP_HIST_MAX * PW < (1ll << 31), "env count too large for 32-bit row offsets");
  ADDHIP_REQUIRE(t->num_tar_steps >= 0 && t->num_tar_steps <= ADDHIP_MAX_TAR_STEPS, "num_tar_steps out of range");
  ADDHIP_REQUIRE(t->num_disc_obs_steps >= 2 && t->num_disc_obs_steps <= ADDHIP_HIST_MAX, "num_disc_obs_steps must be in 2..%d", ADDHIP_HIST_MAX);
  ADDHIP_REQUIRE(t->demo_dt[t->num_disc_obs_steps - 1] == 0.0f, "demo_dt[last] must be 0 (newest demo frame == reference frame)");
  const int hc = t->root_height_obs ? 1 : 0;
  const int vw = t->enable_vel_obs ? 6 + ADDHIP_NUM_DOF : 0;
  ADDHIP_REQUIRE(t->num_phase_encoding >= 0 && t->num_phase_encoding <= MAX_PHASE_ENC, "num_phase_encoding must be in 0..%d", MAX_PHASE_ENC);
  const int want = hc + 6 + ADDHIP_NUM_DOF + vw + (t->enable_phase_obs ? 1 + 2 * t->num_phase_encoding : 0) +
                   t->num_tar_steps * ((hc ? 3 : 2) + 6 + ADDHIP_NUM_DOF);
  ADDHIP_REQUIRE(t->obs_dim == want, "obs_dim %d does not match the task flags (expected %d)", t->obs_dim, want);
  ADDHIP_REQUIRE(t->disc_dim == t->num_disc_obs_steps * (9 + ADDHIP_NUM_DOF + vw), "disc_dim must be %d", t->num_disc_obs_steps * (9 + ADDHIP_NUM_DOF + vw));
  ADDHIP_REQUIRE(!t->enable_vel_obs || e->hist_vel, "hist_vel is required when enable_vel_obs is set");
  ADDHIP_REQUIRE(t->obs_stride >= t->obs_dim && t->disc_stride >= t->disc_dim, "strides smaller than dims");
  const int map_max = map_of(t->enable_vel_obs != 0 || t->num_disc_obs_steps == 4, t->enable_phase_obs != 0);
  ADDHIP_REQUIRE(t->obs_stride + 2 * t->disc_stride <= map_max, "obs_stride + 2*disc_stride must be <= %d", map_max);
  ADDHIP_REQUIRE(e->sim_pose && e->sim_vel && e->time && e->time_off && e->motion_id && e->hist && e->done, "env state pointers missing");
  ADDHIP_REQUIRE(!e->ret_acc || e->len_acc, "ret_acc needs len_acc");
  return 0;
}

inline Tables tables_of(const addhip_motion_t* m) {
  return Tables{m->pose, m->vel, m->clip_start, m->clip_steps, m->total_steps, m->reference_compat, m->dt_inv};
}
inline int env_grid(int num_envs) {
  const int groups = (num_envs + WAVES - 1) / WAVES;
  return groups < 4096 ? groups : 4096;
}

}  // namespace

extern "C" int addhip_env_step(const addhip_motion_t* m, const addhip_task_t* t, const addhip_env_t* e,
                               const addhip_step_out_t* o, int32_t head, void* stream) {
  if (int rc = check_common(m, t, e)) return rc;
  ADDHIP_REQUIRE(o, "null outputs");
  ADDHIP_REQUIRE(!o->obs_timeout || o->obs, "obs_timeout needs obs");
  ADDHIP_REQUIRE(head >= 0 && head < t->num_disc_obs_steps, "head out of range");
  ADDHIP_RECORDABLE(addhip_env_step, m, t, e, o, head);
  hipStream_t st = (hipStream_t)stream;
  // envs per wave: enough waves to fill the chip (256 CUs x ~24 resident waves) first, then amortise the per-group part
  int epw = e->num_envs / 8192;
  epw = epw < 1 ? 1 : (epw > SLOT_MAX ? SLOT_MAX : epw);
  StepArgs a{tables_of(m), e->sim_pose, e->sim_vel, e->time, e->time_off, e->motion_id, e->hist, e->hist_vel, m->clip_len, m->clip_loop,
             e->done, e->contact, e->ret_acc, e->len_acc, e->dof_err_w, e->ref_pose, e->ref_vel, o->obs, o->obs_next_in, o->obs_timeout, o->disc_obs, o->disc_demo,
             o->reward, o->done, o->motion_id_rec, o->motion_time_rec, o->ep_stats, e->num_envs, head, epw};
  const int groups = (e->num_envs + epw - 1) / epw;
  const int blocks = (groups + WAVES - 1) / WAVES;
  const dim3 grid(blocks), block(64 * WAVES);
  // (a four-step task stages its third history / clip row in pass 1: the two-pass instantiation with or without velocity observations)
  const int variant = (t->global_obs ? 4 : 0) | ((t->enable_vel_obs || t->num_disc_obs_steps == 4) ? 2 : 0) | (t->enable_phase_obs ? 1 : 0);
#define ADDHIP_STEP(G, V, P) hipLaunchKernelGGL((env_step_kernel<G, V, P>), grid, block, 0, st, *t, a)
  switch (variant) {
    case 0: ADDHIP_STEP(false, false, false); break;
    case 1: ADDHIP_STEP(false, false, true); break;
    case 2: ADDHIP_STEP(false, true, false); break;
    case 3: ADDHIP_STEP(false, true, true); break;
    case 4: ADDHIP_STEP(true, false, false); break;
    case 5: ADDHIP_STEP(true, false, true); break;
    case 6: ADDHIP_STEP(true, true, false); break;
    default: ADDHIP_STEP(true, true, true); break;
  }
#undef ADDHIP_STEP
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
  ADDHIP_REQUIRE(head >= 0 && head < t->num_disc_obs_steps, "head out of range");
  ADDHIP_RECORDABLE(addhip_env_reset, m, t, e, s, u_clip, u_seg, u_jit, obs_out, disc_obs_out, disc_demo_out, reset_all, head);
  hipStream_t st = (hipStream_t)stream;
  ADDHIP_HIP(hipMemsetAsync(s->temp_bits, 0, sizeof(uint32_t), st));
  hipLaunchKernelGGL(reset_draw_kernel, dim3((e->num_envs + 255) / 256), dim3(256), 0, st, e->num_envs, m->num_clips, e->done, e->motion_id, *s,
                     u_clip, reset_all);
  if (int rc = check_launch("reset_draw_kernel")) return rc;
  ResetArgs a{tables_of(m), m->clip_len, e->sim_pose, e->sim_vel, e->time, e->time_off, e->motion_id, e->hist, e->hist_vel, e->done, e->ref_pose, e->ref_vel,
              u_seg, u_jit, obs_out, disc_obs_out, disc_demo_out, e->num_envs, head, reset_all};
  const dim3 grid(env_grid(e->num_envs)), block(64 * WAVES);
  // (a four-step task stages its third history / clip row in pass 1: the two-pass instantiation with or without velocity observations)
  const int variant = (t->global_obs ? 4 : 0) | ((t->enable_vel_obs || t->num_disc_obs_steps == 4) ? 2 : 0) | (t->enable_phase_obs ? 1 : 0);
#define ADDHIP_RESET(G, V, P) hipLaunchKernelGGL((env_reset_kernel<G, V, P>), grid, block, 0, st, *t, *s, a)
  switch (variant) {
    case 0: ADDHIP_RESET(false, false, false); break;
    case 1: ADDHIP_RESET(false, false, true); break;
    case 2: ADDHIP_RESET(false, true, false); break;
    case 3: ADDHIP_RESET(false, true, true); break;
    case 4: ADDHIP_RESET(true, false, false); break;
    case 5: ADDHIP_RESET(true, false, true); break;
    case 6: ADDHIP_RESET(true, true, false); break;
    default: ADDHIP_RESET(true, true, true); break;
  }
#undef ADDHIP_RESET
  return check_launch("env_reset_kernel");
}

extern "C" int addhip_motion_lookup(const addhip_motion_t* m, const int32_t* ids, const float* times, int32_t count,
                                    int32_t* idx_out, float* pose_out, float* vel_out, void* stream) {
  ADDHIP_REQUIRE(m && ids && times && count > 0, "bad lookup arguments");
  ADDHIP_RECORDABLE(addhip_motion_lookup, m, ids, times, count, idx_out, pose_out, vel_out);
  hipLaunchKernelGGL(lookup_kernel, dim3((count + 3) / 4), dim3(256), 0, (hipStream_t)stream, *m, ids, times, count, idx_out, pose_out, vel_out);
  return check_launch("lookup_kernel");
}

extern "C" int addhip_kin_engine_step(float* sim_pose, float* sim_vel, const float* target, int32_t target_stride, int32_t num_envs,
                                      float lag, float dt, void* stream) {
  ADDHIP_REQUIRE(sim_pose && sim_vel && target && num_envs > 0 && target_stride >= ADDHIP_NUM_DOF, "bad engine-step arguments");
  ADDHIP_RECORDABLE(addhip_kin_engine_step, sim_pose, sim_vel, target, target_stride, num_envs, lag, dt);
  const int n = num_envs * ADDHIP_NUM_DOF;
  hipLaunchKernelGGL(kin_engine_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, sim_pose, sim_vel, target,
                     target_stride, num_envs, lag, dt);
  return check_launch("kin_engine_step_kernel");
}

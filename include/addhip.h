/* addhip.h -- C ABI of the MI355X-native add-gym rollout+PPO hot path (libaddhip.so).
 *
 * The reference (rsamf/add-gym) has no FFI: its hot path is Python/TorchScript
 * (SURVEY.md section 8b).  This header is the boundary a maintainer binds instead of those
 * Python functions; every entry point names the reference function(s) it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (addhip_last_error() has the text);
 *   - all buffers are caller-owned DEVICE pointers (e.g. PyTorch-ROCm allocations), fp32 unless
 *     stated, row-major, rows 16-byte aligned; nothing is allocated or freed by the library;
 *   - every call is asynchronous on the hipStream_t passed as `stream` (void* so that the
 *     header needs no HIP include); no call synchronises or reads device memory on the host,
 *     so all of them can be captured into a hipGraph;
 *   - no global state: pass the same structs each call.
 *
 * Layouts (D = 29 dofs)
 *   pose row  [36] = root_pos xyz | root_rot wxyz | dof_pos[29]
 *   vel  row  [36] = root_vel xyz | root_ang_vel xyz | dof_vel[29] | 0
 *   obs row   [obs_stride >= obs_dim], disc row [disc_stride >= disc_dim]; pad columns are 0.
 */
#ifndef ADDHIP_H
#define ADDHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADDHIP_POSE_W 36
#define ADDHIP_NUM_DOF 29
#define ADDHIP_MAX_TAR_STEPS 8
#define ADDHIP_HIST 3          /* default task.num_disc_obs_steps (configs/task/pose.yaml) */
#define ADDHIP_HIST_MAX 4      /* largest supported */

enum { ADDHIP_DONE_NULL = 0, ADDHIP_DONE_FAIL = 1, ADDHIP_DONE_SUCC = 2, ADDHIP_DONE_TIME = 3 };

const char* addhip_last_error(void);
int addhip_version(void);
/* sizeof() of addhip_motion_t, task_t, env_t, step_out_t, sampler_t, gemm_t, gather_t, rigid_model_t, rigid_dr_t, optimizer_t, section_t, mlp_t,
 * extra_dw_t, mlp_marks_t, ppo_loss_t, ppo_marks_t, disc_loss_t, disc_marks_t, actor_head_t (in that order) -> out[0..18]; returns the number written (19) or -1
 * (count too small).  For bindings to verify their struct layouts against the library they loaded. */
int addhip_abi_sizes(int32_t* out, int32_t count);

/* ---- reference-motion step tables: MotionLib._step_* (anim/motion_lib.py:285-320) ---- */
typedef struct {
  const float* pose;        /* [total_steps,36] */
  const float* vel;         /* [total_steps,36] */
  const int32_t* clip_start; /* [num_clips] row offset added to the frame index.  reference_compat:
                                the RAW-frame start (motion_lib.py:280-282 quirk), else the step start */
  const int32_t* clip_steps; /* [num_clips] rows of each clip (corrected mode clamp) */
  const float* clip_len;    /* [num_clips] seconds (motion_lib.py:202) */
  const int32_t* clip_loop; /* [num_clips] 0 CLAMP / 1 WRAP (anim/motion.py:6-8) */
  int32_t num_clips;
  int32_t total_steps;
  int32_t reference_compat; /* 1: clamp frame to [0,total_steps-1] then add clip_start (motion_lib.py:322-326) */
  float dt_inv;             /* round(1/dt) (motion_lib.py:23) */
} addhip_motion_t;

/* ---- task flags: configs/task/pose.yaml as read by add_observation.py:19-33, add_done.py:21-25,
 *      add_reward.py:12-24 ---- */
typedef struct {
  float dt;                       /* engine.ctrl_dt */
  int32_t global_obs, root_height_obs;
  int32_t num_tar_steps;          /* len(tar_obs_steps) */
  float tar_dt[ADDHIP_MAX_TAR_STEPS]; /* fl32(dt * step_k)   (add_observation.py:215) */
  float demo_dt[ADDHIP_HIST_MAX]; /* fl32(-dt*j) flipped     (add_observation.py:366-369); [0, num_disc_obs_steps) used */
  float max_episode_length;
  int32_t enable_early_termination, pose_termination;
  float pose_termination_dist;
  float pose_w, vel_w, root_pose_w, root_vel_w;
  float pose_scale, vel_scale, root_pose_scale, root_vel_scale;
  int32_t obs_dim, obs_stride, disc_dim, disc_stride;
  int32_t enable_vel_obs;         /* root vel / ang vel / dof vel in the policy and discriminator observations */
  int32_t enable_phase_obs;       /* motion phase (+ 2*num_phase_encoding positional terms) in the policy observation */
  int32_t num_phase_encoding;     /* <= 8 */
  int32_t num_disc_obs_steps;     /* S = 2 .. ADDHIP_HIST_MAX: poses per discriminator observation = depth of the history ring [N, S, 36]
                                     (add_observation.py:276-294, 362-375); demo_dt[0..S) are used, demo_dt[S-1] = 0 */
} addhip_task_t;

/* ---- per-env state between the engine boundary and the agent ---- */
typedef struct {
  int32_t num_envs;
  float* sim_pose;      /* [N,36] simulator state (engine getters, base_engine.py:103-215) */
  float* sim_vel;       /* [N,36] */
  float* time;          /* [N] Environment.time_buf (envs/env.py:124,155) */
  float* time_off;      /* [N] ADDObservation._motion_time_offsets */
  int32_t* motion_id;   /* [N] ADDObservation._motion_ids */
  float* hist;          /* [N,3,36] pose history ring (util/circular_buffer.py), one shared head */
  float* hist_vel;      /* [N,3,36] velocity history ring; required iff task.enable_vel_obs, else NULL */
  int32_t* done;        /* [N] ADDDone.done_buf */
  const uint8_t* contact; /* [N] non-foot ground contact flag (robot.py:221-231) or NULL */
  float* ref_pose;      /* [N,36] ADDObservation.ref_* */
  float* ref_vel;       /* [N,36] */
  /* return tracker (base_agent.py:564-621) */
  float* ret_acc;       /* [N] */
  int32_t* len_acc;     /* [N] */
  const float* dof_err_w; /* [29] per-dof weights of the reward's pose / velocity error sums (task.joint_err_w expanded to
                             dofs, add_reward.py:28-52) or NULL = all ones; the done flags use the unweighted mean */
} addhip_env_t;

/* outputs of one env step; any pointer may be NULL to skip that output */
typedef struct {
  float* obs;         /* [N,obs_stride]  next_obs slot of the experience buffer */
  float* obs_next_in; /* [N,obs_stride]  optional second copy of the same rows (e.g. a separate next_obs buffer) or NULL */
  float* obs_timeout; /* [N,obs_stride]  row n receives a copy of the obs row when done[n]==TIME (the one case where
                         the critic needs the pre-reset next obs, ppo_agent.py:117-133) or NULL */
  float* disc_obs;    /* [N,disc_stride] */
  float* disc_demo;   /* [N,disc_stride] */
  float* reward;      /* [N] */
  int32_t* done;      /* [N] experience-buffer slot (copy of env.done) */
  int32_t* motion_id_rec;   /* [N] */
  float* motion_time_rec;   /* [N] */
  float* ep_stats;    /* [3] += (sum of finished returns, sum of finished lengths, #finished) or NULL */
} addhip_step_out_t;

/* ADDAgent._step_env after scene.step() (add_agent.py:204-219): time += dt (env.py:155);
 * _update_ref_motion + _update_disc_hist (add_observation.py:163-207); compute_add_obs /
 * compute_disc_obs x2 (add_observation.py:231-306, 356-419, 422-717); compute_reward
 * (add_reward.py:103-177); compute_done (add_done.py:96-147); ReturnTracker.update
 * (base_agent.py:596-621).  `head` = ring slot that receives the new state.
 * One launch (env_step_kernel): observations, ring push, reference rows, reward, done flags and return tracker.
 * task.enable_vel_obs adds the velocity terms of compute_char_obs / compute_vel_obs
 * (add_observation.py:445-452, 502-517; needs e->hist_vel); task.enable_phase_obs adds
 * compute_phase_obs (:557-575). */
int addhip_env_step(const addhip_motion_t* m, const addhip_task_t* t, const addhip_env_t* e,
                    const addhip_step_out_t* o, int32_t head, void* stream);

/* adaptive start-time sampler state: AdaptiveSegmentSampler (learning/sampler.py) */
typedef struct {
  float* errors;          /* [num_clips,num_segments] */
  const float* seg_size;  /* [num_clips] clip_len/num_segments (sampler.py:13-15) */
  const float* clip_cdf;  /* [num_clips] inclusive CDF of the clip weights (motion_lib.py:35-39) */
  int32_t num_segments;
  float temperature;      /* <=0: max(err over the clips drawn in this batch)+1e-6 (sampler.py:66-69) */
  float min_start_time;   /* (num_disc_obs_steps-1)*dt (add_motion.py:31) */
  int32_t rand_reset;
  /* scratch (device): */
  uint32_t* temp_bits;    /* [1] running max of the drawn clips' errors (as float bits) */
  float* err_sum;         /* [num_clips*num_segments] */
  float* err_cnt;         /* [num_clips*num_segments] */
} addhip_sampler_t;

/* ADDAgent._reset_envs (add_agent.py:221-233) for every env whose done flag != NULL (or all envs
 * when reset_all): ADDMotion.sample_time (add_motion.py:53-61, sampler.py:57-92), time=0, done=NULL,
 * ref state -> simulator (add_observation.py:308-332), history refill (:334-344), observations
 * recomputed for the reset envs.  u_clip/u_seg/u_jit: [N] uniforms in [0,1) (one draw per env;
 * consumed only where a reset happens).  `head` = slot of the OLDEST ring entry after the push.
 * Two launches (clip draw + batch temperature, then the rest). */
int addhip_env_reset(const addhip_motion_t* m, const addhip_task_t* t, const addhip_env_t* e,
                     const addhip_sampler_t* s, const float* u_clip, const float* u_seg,
                     const float* u_jit, float* obs_out, float* disc_obs_out, float* disc_demo_out,
                     int32_t reset_all, int32_t head, void* stream);

/* lookup only: MotionLib.get_precomputed_motion_step (motion_lib.py:322-335) for M (id,time) pairs */
int addhip_motion_lookup(const addhip_motion_t* m, const int32_t* ids, const float* times, int32_t count,
                         int32_t* idx_out, float* pose_out, float* vel_out, void* stream);

/* KinematicEngine.step (build-defined stand-in simulator behind base_engine.BaseEntity;
 * control_dofs_position + scene.step): dof += lag*(target-dof), dof_vel = delta/dt */
int addhip_kin_engine_step(float* sim_pose, float* sim_vel, const float* target, int32_t target_stride,
                           int32_t num_envs, float lag, float dt, void* stream);

/* ---- rigid-body engine behind the plugin API: BaseScene.step() + BaseEntity.control_dofs_position / get_contacts
 *      (engine/base_engine.py:93-455; the reference delegates to Genesis or MuJoCo-Warp: engine/mjwarp_engine.py:807-851 PD
 *      torque loop, :896-986 contacts, :1554-1611 substepped step).  Articulated-body algorithm over the kinematic tree with a
 *      floating base; joint PD (stable-PD form), joint damping / armature, and penalty ground contacts of per-body collision
 *      spheres are integrated linearly-implicitly; semi-implicit Euler.  Build-defined dynamics (parity is against
 *      oracle/rigid.py and physical invariants, not against the reference's simulators). ---- */
#define ADDHIP_RIGID_BODY_W 32
#define ADDHIP_RIGID_TOPO_W 8
typedef struct {
  int32_t num_bodies;      /* root + one hinge body per dof (== ADDHIP_NUM_DOF + 1 for the packed state rows) */
  int32_t num_points;
  /* per body, in TRAVERSAL order (depth-first pre-order: a single child directly follows its parent), ADDHIP_RIGID_BODY_W floats:
   *   0-2 offset in the parent frame | 3-11 child->parent rotation at q=0 (row major) | 12 mass | 13-15 mass*com |
   *   16-21 inertia about the body origin xx xy xz yy yz zz | 22 lo 23 hi | 24 damping 25 armature 26 |torque| limit | 27 kp 28 kv |
   *   29 radius of the sphere about the body origin that holds all its collision spheres (0 if none) */
  const float* body;
  /* per body, traversal order, ADDHIP_RIGID_TOPO_W ints: parent (traversal index) | hinge axis 0/1/2 (x/y/z of the body frame) |
   *   dof column (breadth-first: column 7+dof of the pose row) | number of children | branch accumulator slot (-1 unless the
   *   body has more than one child; at most 4 branch bodies) | first collision point | number of points | link index
   *   (breadth-first body index, bit position in contact_bits) */
  const int32_t* topo;
  const float* points;     /* [num_points,4] x y z radius in the body frame, grouped by body */
  float dt;                /* control step */
  int32_t substeps;
  float gravity;           /* 9.81 */
  float contact_stiffness, contact_damping, friction, friction_vel_eps;
  float limit_stiffness, max_torque, limit_margin;
  uint32_t termination_mask; /* bit b: a ground contact of link b raises contact_flag (robot.py:221-231 with add_done.py:36-45) */
  /* domain randomisation (build-defined extension; the reference has none): per-env multipliers [N,2] = (PD gain scale applied to kp
   * and kv, ground friction coefficient replacing `friction`), or NULL */
  const float* env_scale;
  /* optional (NULL = one lane per environment): the depth-first body order cut into at most four chains (maximal runs whose
   * parent is the previous body), one lane of a quad each: int32 [4][16] = len, start step, attach lane (-1 = the root),
   * body indices; a chain hanging off body j of another chain starts at that body's step + 1; max(start + len) <= 10 */
  const int32_t* chains;
} addhip_rigid_model_t;
/* One control step for every env, in place on the packed state rows (pose[N,36], vel[N,36]); target [N,target_stride] = joint
 * position targets (breadth-first dof order).  contact_flag [N] (or NULL): 1 if a link selected by termination_mask touched
 * the ground in the last substep; contact_bits [N] (or NULL): one bit per link. */
int addhip_rigid_step(const addhip_rigid_model_t* m, float* sim_pose, float* sim_vel, const float* target, int32_t target_stride,
                      int32_t num_envs, uint8_t* contact_flag, uint32_t* contact_bits, void* stream);

/* Domain randomisation of the rigid-body engine -- a BUILD-DEFINED EXTENSION (add-gym has no domain randomisation: SURVEY.md
 * section 0; BASELINE configs[4] asks for it inside the hipGraph-captured rollout step).  One launch per control step, before
 * addhip_rigid_step, driven by a DEVICE-resident counter, so a captured rollout replays it unchanged and draws what the
 * call-by-call run draws.  step_counter = uint64[2] in device memory: [0] the control-step index s (0 at build), [1] scratch, 0
 * between launches.  advance = 1: when s > 0 and s is a multiple of resample_interval (> 0), env_scale[N,2] = (PD gain scale, ground
 * friction) is redrawn uniformly from the ranges (Philox stream (8<<40)+s under `seed`); when s > 0 and s is a multiple of
 * push_interval (> 0), sim_vel[:,0:2] += U(-push_velocity, push_velocity) (stream (9<<40)+s); then step_counter[0] = s + 1.
 * advance = 0 (initialisation): env_scale is drawn for step index s unconditionally, the counter stays. */
typedef struct {
  uint64_t seed;
  int32_t resample_interval, push_interval;
  float gain_lo, gain_hi, friction_lo, friction_hi, push_velocity;
} addhip_rigid_dr_t;
int addhip_rigid_randomize(const addhip_rigid_dr_t* dr, float* env_scale, float* sim_vel, int32_t num_envs, uint64_t* step_counter,
                           int32_t advance, void* stream);

/* ---- dense layers: torch.nn.Linear(+ReLU) stacks of PPOModel/ADDModel (ppo_model.py:13-21,
 *      add_model.py:12-15, nets/fc_*layers_1024units.py) on fp32 MFMA ---- */
enum { ADDHIP_EPI_NONE = 0, ADDHIP_EPI_BIAS = 1, ADDHIP_EPI_BIAS_RELU = 2, ADDHIP_EPI_MASK = 3 };
/* ADDHIP_PREC_F32: v_mfma_f32_32x32x2_f32.  ADDHIP_PREC_BF16X3: each fp32 operand split exactly into three bf16 chunks,
 * six bf16 MFMAs per k-step accumulated in fp32 (error bound 2^-23 |a||b| per product, i.e. fp32-level; the reference
 * itself runs its matmuls in TF32, main.py:16-18).  ADDHIP_PREC_BF16X2: the two leading chunks only (16 significant bits per operand, three MFMAs per
 * k-step: error bound 2^-15 |a||b|, 64x below TF32).  ADDHIP_PREC_BF16: operands truncated to bf16, fp32 accumulate.
 * The split paths are used for shapes that fill the chip with 128x128 tiles; other shapes always take the fp32 path. */
enum { ADDHIP_PREC_F32 = 0, ADDHIP_PREC_BF16 = 1, ADDHIP_PREC_BF16X2 = 2, ADDHIP_PREC_BF16X3 = 3, ADDHIP_PREC_F16X2 = 4 };
/* ADDHIP_PREC_F16X2: each fp32 operand value, scaled by an exact power of two chosen per TENSOR, is split into two fp16 values
 * hi = fp16(x s), lo = fp16(x s - hi) (round to nearest: 22 significant bits + sign of the residual) and all FOUR fp16 products are
 * accumulated in fp32 by v_mfma_f32_32x32x16_f16: operand representation error <= 2^-22 |x| each, i.e. a bound of 2^-21 |a||b| per product
 * (fp32 MFMA: 0 per product + 2^-24 per accumulation; measured at K = 1024: both 3.3e-7 of sum |a||b|), at 4 instead of 6 matrix
 * instructions per k-step -- the chip sustains 395-435 TFLOP/s of fp32 products on this pattern against 280-290 on the bf16x3 one
 * (tools/ubench/mfma_x3_variants.hip).  fp16 has 5 exponent bits, so the scale must put the tensor's largest magnitude just below
 * 2^15: the descriptor carries, per operand, a device array of ADDHIP_AMAX_SLOTS float bit patterns whose maximum is an upper bound of
 * max |x| over the operand (a_amax, b_amax), written by the operand's producer -- every kernel of this library that produces a GEMM
 * operand can track it (addhip_gemm_t.amax_out, the `amax` arguments of the loss-head kernels, addhip_amax_f32 for anything else).
 * A descriptor with precision F16X2 but without both arrays runs as ADDHIP_PREC_BF16X3.  Values below 2^-18 of the tensor's maximum
 * keep an ABSOLUTE error of 2^-40 of that maximum. */
#define ADDHIP_AMAX_SLOTS 64
/* ---- plane storage.  16-bit storage formats of fp32 values, for GEMM operands kept in HBM (addhip_gemm_t.operands_bf16, .c16_planes and
 *      the `planes16` argument / field of every producer of such operands):
 *   ADDHIP_STORE_BF16   (1)  one bf16 per value, rounded to nearest even (agent.matmul_precision = bf16: 8 significant bits);
 *   ADDHIP_STORE_BF16X3 (3)  three bf16 per value, hi + mid + lo == x EXACTLY (split by truncation), interleaved in groups of 8 values:
 *                            element c of plane p of a row of ld values (ld % 8 == 0) is the uint16 at (c / 8) * 24 + p * 8 + c % 8, row r
 *                            starts at uint16 index 3 * r * ld (leading dimensions always count VALUES).  A GEMM on such operands forms the
 *                            six bf16 products above 2^-24 |a||b| (csrc/gemm_x3.hip): the error bound of ADDHIP_PREC_BF16X3, i.e. of the fp32
 *                            MFMA, without splitting operands inside the GEMM (agent.matmul_precision = bf16x3, update step). */
enum { ADDHIP_STORE_BF16 = 1, ADDHIP_STORE_BF16X3 = 3 };
typedef struct {
  int32_t M, N, K;          /* C[M,N] = sum_k A(m,k) * B(n,k) */
  const float* A; int32_t lda; int32_t a_kcontig; /* 1: A[m*lda+k], 0: A[k*lda+m] */
  const float* B; int32_t ldb; int32_t b_kcontig; /* 1: B[n*ldb+k], 0: B[k*ldb+n] */
  float* C; int32_t ldc;
  int32_t epilogue;         /* ADDHIP_EPI_* */
  const float* bias;        /* [N] for BIAS / BIAS_RELU */
  const float* mask; int32_t ldmask; /* MASK: C = acc * (mask[m,n] > 0) */
  const float* a_mean; const float* a_std; /* optional fused (a-mean)/std on A when a_kcontig (Normalizer.normalize, normalizer.py:107-110) */
  int32_t split_k;          /* >1: C is [split_k, M, ldc] partial slabs (slab stride M*ldc) */
  float alpha;              /* scale applied to acc before the epilogue */
  float* colsum;            /* optional, MASK epilogue only: colsum[n] += sum_m C[m,n] (the bias gradient of the layer
                               whose pre-activation gradient this GEMM produces; atomics, caller zeroes) */
  int32_t precision;        /* ADDHIP_PREC_*: how the fp32 products are formed (operands and results are fp32 either way) */
  /* ReLU sign bits, 1 bit per element instead of re-reading the fp32 activations as the mask of the backward pass:
   * word [m*ldbits + n/32], bit n%32 = (C[m,n] > 0).  relu_bits: written by the BIAS_RELU epilogue (optional).
   * mask_bits: read by the MASK epilogue instead of `mask` (optional; M > 8 only).  ldbits >= ceil(N/32). */
  uint32_t* relu_bits;
  const uint32_t* mask_bits;
  int32_t ldbits;
  /* split_k > 1 with accumulate != 0: instead of writing split_k partial slabs, every K slice ADDS its partial product into the one
   * C[M,ldc] with hardware fp32 atomics (no epilogue; the caller zeroes C, e.g. the flat gradient buffer at the start of an optimiser
   * step; several GEMMs may accumulate into the same C).  Summation order is not fixed: results vary in the last bits run to run. */
  int32_t accumulate;
  /* bf16 STORAGE (agent.matmul_precision = bf16).  operands_bf16 != 0: A and B point at bf16 values (lda / ldb count bf16 elements; K or the
   * contiguous extent, and the leading dimensions, multiples of 8), products by v_mfma_f32_32x32x16_bf16 with fp32 accumulation, no
   * conversion anywhere.  Epilogues, sign bits, column sums and split-K slabs (fp32) as for fp32 operands; no fused normalisation.
   * C16 (optional, with fp32 operands too): a second copy of the result rounded to bf16 (nearest even), leading dimension ldc16; with
   * it C may be NULL (not with split-K slabs, which are fp32). */
  int32_t operands_bf16;    /* 0 | ADDHIP_STORE_BF16 | ADDHIP_STORE_BF16X3 (A and B in plane storage; lda / ldb count values, multiples of 8) */
  uint16_t* C16;
  int32_t ldc16;
  /* 0 (always, in product code): the dispatcher picks the kernel configuration from the shape.  ADDHIP_GEMM_HINT_* bits override one
   * of its choices for measurements and tests (tools/gemm_*_one.py, tests/test_hip_gemm.py); the result is the same either way.
   * No process environment variable influences kernel selection. */
  int32_t hint;
  /* colsum_replicas R > 1: `colsum` points at R rows of ldcs floats and the column sums of a 32-row block of C are added to row
   * (block index % R).  All row tiles of a launch add to the same N floats otherwise, and same-line float atomics serialise in the memory
   * system (~45 ns each: 6-8 us of a 16384-row launch); R = 16 removes that.  The rows are summed -- and cleared -- by
   * addhip_slab_reduce_pair.  0 / 1: plain colsum[N]. */
  int32_t colsum_replicas;
  int32_t ldcs;
  /* format of C16: 0 / ADDHIP_STORE_BF16 = rounded to bf16; ADDHIP_STORE_BF16X3 = the exact three-plane split of the fp32 result (N and
   * ldc16 multiples of 8), e.g. the hidden activations / pre-activation gradients the next plane-storage GEMM reads */
  int32_t c16_planes;
  /* ADDHIP_PREC_F16X2 (above): upper bounds of max |A|, max |B| as ADDHIP_AMAX_SLOTS float bit patterns each (the maximum over the slots
   * counts; zero slots are fine), read at kernel start; NULL (either): the launch runs as ADDHIP_PREC_BF16X3. */
  const uint32_t* a_amax; const uint32_t* b_amax;
  /* optional, any precision: max |C| over this launch's results (after the epilogue), atomically maxed as float bits into slot
   * (workgroup index % ADDHIP_AMAX_SLOTS) of this array, which the caller zeroes (addhip_fill_zero) before the launch(es) that produce
   * the tensor; not with split-K slabs */
  uint32_t* amax_out;
} addhip_gemm_t;
enum {
  ADDHIP_GEMM_HINT_BIG_TILE = 1,     /* bf16 operands: the 256x256 ring kernel on eligible shapes (M, N multiples of 256, K of 64) */
  ADDHIP_GEMM_HINT_NO_BIG_TILE = 2,  /* bf16 operands: never the 256x256 kernel */
  ADDHIP_GEMM_HINT_ONE_STAGE = 4,    /* 128x128 tiles: one LDS stage x 3-4 workgroups per CU */
  ADDHIP_GEMM_HINT_TWO_STAGE = 8,    /* 128x128 tiles: two LDS stages x 2 workgroups per CU */
  ADDHIP_GEMM_HINT_REG_STAGED = 16,  /* fp32 operands: the register-staged tile kernel instead of the LDS-DMA one */
  ADDHIP_GEMM_HINT_WIDE_TILE = 32    /* plane-stored operands: the 256x128 configuration (BIG_TILE: 256x256, NO_BIG_TILE: 128x128) */
};
int addhip_gemm_f32(const addhip_gemm_t* g, void* stream);
/* Up to ADDHIP_GEMM_MAX_GROUP problems of the SAME shape, operand layouts, epilogue kind, split, precision and storage (different
 * buffers) as ONE launch -- e.g. the actor's and the critic's equal-shaped layers of an update step (ppo_agent.py:194-275 runs them
 * one after the other).  A launch then has several rounds of tiles, so one round's write-out overlaps the next round's first
 * loads instead of the whole chip writing in lock-step.  Results are those of `count` addhip_gemm_f32 calls; shapes that do not
 * take the 128x128 tile kernel are launched one by one. */
#define ADDHIP_GEMM_MAX_GROUP 4
int addhip_gemm_grouped(const addhip_gemm_t* problems, int32_t count, void* stream);

/* dst[r*ld_dst + c] = bf16(src[r*ld_src + c]), round to nearest even (bf16-storage mode: minibatch inputs, head gradients, the weight
 * shadow after an optimiser step); cols and both leading dimensions multiples of 4 */
int addhip_to_bf16(const float* src, uint16_t* dst, int64_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, void* stream);
/* dst = bf16((src - mean[c]) / std[c]), round to nearest even: Normalizer.normalize (normalizer.py:107-110) into the bf16 rows the
 * bf16-storage GEMMs read (the rollout / evaluation passes of agent.rollout_precision = bf16_storage); same size rules */
int addhip_normalize_to_bf16(const float* src, const float* mean, const float* std, uint16_t* dst, int64_t rows, int32_t cols, int32_t ld_src,
                             int32_t ld_dst, void* stream);
/* the same into plane storage (ADDHIP_STORE_BF16X3): cols and ld_dst multiples of 8, ld_src of 4 */
int addhip_to_bf16x3(const float* src, uint16_t* dst, int64_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, void* stream);
/* transposing variant: dst[c*ld_dst + r] = bf16(src[r*ld_src + c]) -- the [in,out] copy of a weight matrix [out,in], which lets the
 * backward (dX) GEMMs of the bf16-storage mode read the weights k-contiguously like the forward ones */
int addhip_to_bf16_t(const float* src, uint16_t* dst, int32_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, void* stream);
/* The whole bf16 weight shadow after an optimiser step in ONE launch: flat16[i] = bf16(params[i]) for i < count, and for each of the
 * n_mats (<= ADDHIP_SHADOW_MAX_MATS) row-major matrices [rows, cols] at params + offset the transposed copy [cols, rows] at
 * trans16 + offset (offset, rows, cols: host arrays).  flat16 may be NULL (the flat shadow is then written by addhip_optimizer_step and only
 * the transposed copies of the listed matrices are refreshed -- e.g. one net's, on that net's stream, between its forward and backward pass) */
#define ADDHIP_SHADOW_MAX_MATS 8
int addhip_shadow_refresh(const float* params, uint16_t* flat16, uint16_t* trans16, int64_t count, const int64_t* offset, const int32_t* rows,
                          const int32_t* cols, int32_t n_mats, int32_t planes16 /* ADDHIP_STORE_*: format of flat16 and trans16; X3: the flat
                          buffer is split in groups of 8 along its flat index (every tensor offset, row length and count % 8 == 0), the
                          transposed copies row by row; both buffers hold 3 * count uint16 */, void* stream);

/* slots[block % ADDHIP_AMAX_SLOTS] = max(slots[..], max |x[i]|) as float bit patterns, for i < count (caller zeroes slots): the operand
 * bound ADDHIP_PREC_F16X2 GEMMs read (a_amax / b_amax) for tensors no kernel of this library produced -- e.g. the flat parameter
 * buffer after an optimiser step or a checkpoint load */
int addhip_amax_f32(const float* x, int64_t count, uint32_t* slots, void* stream);

/* out[n] (+)= scale * sum over `slabs` of in[s*slab_stride + n]  (split-K combine, grads) */
int addhip_slab_reduce(const float* in, int32_t slabs, int64_t slab_stride, float* out, int64_t count,
                       float scale, int32_t accumulate, void* stream);
/* Two reductions in one launch: the first exactly addhip_slab_reduce; the second  out2[n] (+)= sum over rows2 rows of in2[r*ld2 + n]  for
 * n < count2, after which (clear2 != 0) the rows of in2 are zeroed -- the replicated bias-gradient column sums of a dX GEMM
 * (addhip_gemm_t.colsum_replicas), folded into the split-K combine of the layer below that follows it in every backward pass. */
int addhip_slab_reduce_pair(const float* in, int32_t slabs, int64_t slab_stride, float* out, int64_t count, float scale, int32_t accumulate,
                            float* in2, int32_t rows2, int32_t ld2, float* out2, int32_t count2, int32_t accumulate2, int32_t clear2, void* stream);
/* out[n] (+)= scale * sum_m X[m*ld+n]   (bias gradients) */
int addhip_col_sum(const float* X, int32_t M, int32_t N, int32_t ld, float* out, float scale,
                   int32_t accumulate, void* stream);
/* The same with a FIXED summation order (no float atomics: the result is bit-identical from run to run): ADDHIP_ORDERED_BLOCKS row slices
 * are summed into scratch[slice][n] (scratch: >= ADDHIP_ORDERED_BLOCKS * N floats), then added slice by slice.  Two launches. */
#define ADDHIP_ORDERED_BLOCKS 64
int addhip_col_sum_ordered(const float* X, int32_t M, int32_t N, int32_t ld, float* out, float scale, int32_t accumulate, float* scratch, void* stream);

/* ---- rollout actor head: DistributionGaussianDiag.sample/log_prob (distribution_gaussian_diag.py:84-94)
 *      + Normalizer.unnormalize (normalizer.py:112-114) + exp-buffer record (ppo_agent.py:72-109).
 *      deterministic != 0: mode for every env (test mode).  Otherwise, with explore_u [N] uniforms in [0,1) (or NULL =
 *      always explore): env n samples iff explore_u[n] < exp_prob, else takes the mode and gets rand_mask 0
 *      (rand_action_mask = bernoulli(exp_prob), ppo_agent.py:80-88, 161-168) ---- */
/* The policy's standard deviation.  actor_std_type FIXED (distribution_gaussian_diag.py:26-31): the scalar arguments `std`, `logp_const`
 * of the calls below, `dist` = NULL.  actor_std_type CONSTANT (:32-37, a trainable log-std per action dimension): `dist` points at
 * ADDHIP_DIST_FLOATS device floats kept current by addhip_dist_refresh -- [0, 29) std_j = exp(logstd_j), [32] the log-probability
 * constant fl32(-0.5*29*log(2pi)) - sum(logstd), [33] the entropy sum(logstd) + 0.5*29*log(2 pi e) -- and the scalars are ignored. */
#define ADDHIP_DIST_FLOATS 64
int addhip_dist_refresh(const float* logstd /*[29]*/, float* dist /*[ADDHIP_DIST_FLOATS]*/, void* stream);
int addhip_actor_sample(const float* mean, int32_t ld_mean, const float* noise /*[N,29] N(0,1)*/, float std,
                        float logp_const /* fl32(-0.5*29*log(2pi)) - sum(logstd) */, const float* dist /* or NULL */,
                        const float* logstd_rows /* or NULL; actor_std_type VARIABLE: [N, ld_mean] per-sample log-std (the second head's columns) */,
                        const float* a_mean, const float* a_std, int32_t num_envs,
                        int32_t deterministic, const float* explore_u, float exp_prob, float* action /*[N,32]*/, float* a_logp,
                        float* rand_mask, void* stream);

/* counter-based Philox4x32-10 fills (stateless: element i of call (seed,stream_id) is fixed) */
int addhip_fill_normal(float* out, int64_t count, uint64_t seed, uint64_t stream_id, void* stream_);
int addhip_fill_uniform(float* out, int64_t count, uint64_t seed, uint64_t stream_id, void* stream_);
/* same, with the stream id = stream_id + stream_base[0] read ON THE DEVICE at execution time: a captured hipGraph (the whole rollout
 * of an iteration) then draws fresh numbers on every replay once the caller has advanced the counter */
int addhip_fill_normal_at(float* out, int64_t count, uint64_t seed, uint64_t stream_id, const uint64_t* stream_base, void* stream_);
int addhip_fill_uniform_at(float* out, int64_t count, uint64_t seed, uint64_t stream_id, const uint64_t* stream_base, void* stream_);

/* ---- build-train-data (add_agent.py:110-139, amp_agent.py:194-206, sampler.py:20-55) ---- */
/* norm_diff = (demo-agent)/max(mean_abs,1e-4); err = sum((agent-demo)^2) scatter-added per (clip,segment);
 * also sum|demo-agent| per column into abs_sum (DiffNormalizer.record, diff_normalizer.py:24-31) */
int addhip_disc_prep(const float* disc_obs, const float* disc_demo, int32_t stride, int32_t dim, int64_t rows,
                     const float* mean_abs, float min_diff, float* norm_diff,
                     const int32_t* motion_id, const float* motion_time, const addhip_sampler_t* s,
                     int32_t num_clips, float* abs_sum, void* stream);
/* errors = where(cnt>0, 0.9*errors + 0.1*sum/cnt, errors); clears sum/cnt (sampler.py:43-55) */
int addhip_sampler_update(const addhip_sampler_t* s, int32_t num_clips, void* stream);
/* r = task_w*task_r + disc_w * (-log(max(1-sigmoid(logit),1e-4))*scale); stats[0..1] += (sum, sumsq) of disc_r */
int addhip_disc_reward(const float* logits, float* reward_inout, int64_t count, float scale, float task_w,
                       float disc_w, float* stats, void* stream);

/* row-wise GEMV head: out[m] = dot(H[m,:K], w) + b   (critic_out / disc_logits, ppo_model.py:18-21) */
int addhip_head_gemv(const float* H, int32_t ld, int32_t K, int64_t rows, const float* w, const float* b,
                     float* out, void* stream);

/* PPOAgent._build_train_data (ppo_agent.py:111-159) + compute_td_lambda_return (base_agent.py:624-647):
 * next value = succ/fail value where done in {SUCC,FAIL}, else next_vals; reverse scan; adv = ret - vals; then
 * mean/std (unbiased) over samples with rand_mask==1, normalise, clamp.  next_vals [T,N] is read-only and may alias
 * vals shifted by one step (V(next_obs[t]) == V(obs[t+1]) wherever no reset happened); timeout_vals [N] (or NULL)
 * then supplies V(true next obs) for the samples with done==TIME, whose obs[t+1] row already holds the reset obs
 * (at most one per env per call: requires max_episode_length >= T*dt).
 * scratch: [>= 4 + 2*1024] floats. stats_out[0..1]=mean,std */
int addhip_td_lambda_adv(const float* reward, const float* next_vals, const float* timeout_vals, const float* vals, const int32_t* done,
                         const float* rand_mask, int32_t T, int32_t N, float discount, float td_lambda,
                         float succ_val, float fail_val, float adv_clip, float* tar_val, float* adv,
                         float* scratch, float* stats_out, void* stream);

/* hipMemsetAsync(p, 0, 4*count): e.g. the flat gradient buffer once per optimiser step (MPOptimizer.step's zero_grad,
 * mp_optimizer.py:14-16), so that bias gradients can be accumulated by atomics from several kernels */
int addhip_fill_zero(float* p, int64_t count, void* stream);

/* ---- normalisers ---- */
/* sum[n] += sum_m x[m,n]; sumsq[n] += sum_m x^2   (Normalizer.record, normalizer.py:25-35) */
int addhip_norm_accum(const float* X, int64_t rows, int32_t dim, int32_t ld, float* sum, float* sumsq, void* stream);
/* Normalizer.update (normalizer.py:37-80); count is int64 on device; new_count passed by value */
int addhip_norm_merge(float* mean, float* std, float* mean_sq, int64_t* count, float* sum, float* sumsq,
                      int64_t new_count, int32_t dim, float min_var, int32_t first, void* stream);
/* DiffNormalizer.update (diff_normalizer.py:33-45) */
int addhip_diffnorm_merge(float* mean_abs, int64_t* count, float* abs_sum, int64_t new_count, int32_t dim, void* stream);

/* ---- minibatch assembly: ExperienceBuffer.sample (experience_buffer.py:74-82) fused with the
 *      normalisations of _compute_loss (ppo_agent.py:194-196, add_agent.py:152-153) ---- */
typedef struct {
  const int64_t* idx; int32_t count;     /* rows into the flat [T*N] buffers */
  const float* obs; int32_t obs_stride; int32_t obs_dim; const float* obs_mean; const float* obs_std;
  const float* action; const float* a_mean; const float* a_std; /* action [.,32] */
  const float* a_logp; const float* adv; const float* tar_val; const float* rand_mask;
  const float* disc_obs; const float* disc_demo; int32_t disc_stride; int32_t disc_dim;
  const float* mean_abs; float min_diff;
  float* norm_obs; float* norm_action; float* o_logp; float* o_adv; float* o_tar_val; float* o_mask; float* norm_diff;
  uint16_t* norm_obs16; uint16_t* norm_diff16; /* optional (NULL = none): bf16 copies of norm_obs / norm_diff, same strides (bf16-storage mode) */
  int32_t planes16;                            /* ADDHIP_STORE_* format of those two (0 = ADDHIP_STORE_BF16) */
  uint32_t* obs_amax; uint32_t* diff_amax;     /* optional: max |norm_obs|, max |norm_diff| tracked into ADDHIP_AMAX_SLOTS slots each (caller zeroes; ADDHIP_PREC_F16X2) */
} addhip_gather_t;
int addhip_gather_minibatch(const addhip_gather_t* g, void* stream);

/* ---- losses (forward value + gradient w.r.t. the head outputs) ---- */
/* PPOAgent._compute_actor_loss (ppo_agent.py:221-275) + _compute_action_bound_loss (base_agent.py:522-546) + the mean
 * regulariser reg_weight * mean(sum mean^2) (action_reg_weight, ppo_agent.py:268-272, distribution_gaussian_diag.py:113-116; the
 * entropy term of :262-266 is a constant for the fixed-std policy and has no gradient).
 * d_mean[M,32] = d loss / d mean;  stats[0..3] += means over the exploring samples of {min-term, clipped?, ratio, bound},
 * stats[5] += mean of sum mean^2 (only when reg_weight != 0).
 * With `dist` (trainable log-std): g_logstd[j] += d loss / d logstd_j = sum_rows d loss/d logp * (((a_j - mean_j) / std_j)^2 - 1)
 * (float atomics; :90-94 differentiated). */
int addhip_actor_loss(const float* mean, const float* norm_action, const float* old_logp, const float* adv,
                      const float* rand_mask, int32_t M, float std, float logp_const, const float* dist /* or NULL */, float clip_ratio,
                      float bound_weight, float reg_weight, float loss_scale, const float* n_valid /*device [1]*/, float* d_mean,
                      float* g_logstd /* [29], with dist */, float* stats,
                      int32_t ld_mean /* row stride of mean and d_mean: 32, or 64 with logstd_rows */,
                      const float* logstd_rows /* or NULL; VARIABLE: per-sample log-std [M, ld_mean] (usually mean + 32); d loss / d logstd then goes
                                                  to columns 32..63 of d_mean: the two heads are one 64-wide product */,
                      float entropy_weight /* with logstd_rows: the entropy bonus -w * mean(entropy) (ppo_agent.py:262-266) -- its gradient on every
                                              log-std, stats[6] += mean entropy over the exploring samples */, void* stream);
/* The actor's head section as ONE launch: mean = H Wh^T + bh (DistributionGaussianDiagBuilder.forward, distribution_gaussian_diag.py:47-58),
 * addhip_actor_loss on it (same arguments, same stats slots), and the backward step through the head into the last hidden layer:
 *   dz = (d_mean Wh) * (H > 0)   -> dz [rows, hidden] fp32 and / or dz16 (ADDHIP_STORE_* format planes16),
 *   db_top[k] += sum_rows dz     -> added to row (workgroup % gb_replicas) of gb_top [gb_replicas, ld_gb] (gb_replicas = 1: plain atomics on one row),
 *   (dWh | dbh | dlogstd) partial sums -> slabs [num_slabs][ADDHIP_ACTOR_HEAD_SLAB(hidden) = 32 * hidden + 64], one per workgroup, to be
 *                                   combined by addhip_slab_reduce(slabs, num_slabs, 32 * hidden + 64, gWh, 32 * hidden + 32, 1, 0) when the
 *                                   head's bias gradient directly follows its weight gradient in memory (or by two calls); with `dist`
 *                                   the last 32 floats of a slab are the log-std gradient's partial sums (zeros without).
 * Replaces three 32-wide GEMM launches + addhip_actor_loss + addhip_col_sum of the step (csrc/actor_head.hip).  hidden: 128, 256 or 512;
 * Wh [32, hidden] with zero rows past ADDHIP_NUM_DOF; num_slabs = addhip_actor_head_slabs(rows). */
typedef struct {
  int32_t rows, hidden;
  const float* H; const float* Wh; const float* bh;
  const float* norm_action; const float* old_logp; const float* adv; const float* rand_mask; const float* n_valid;
  float action_std, logp_const, clip_ratio, bound_weight, reg_weight, loss_scale;
  const float* dist;        /* trainable log-std (addhip_dist_refresh) or NULL: the scalars above */
  float* dz; uint16_t* dz16; int32_t planes16;
  float* slabs; int32_t num_slabs;
  float* gb_top; int32_t gb_replicas, ld_gb;
  float* stats;
  uint32_t* amax;           /* optional: max |dz| into ADDHIP_AMAX_SLOTS slots (caller zeroes) */
} addhip_actor_head_t;
#define ADDHIP_ACTOR_HEAD_SLAB(hidden) (32 * (hidden) + 64)
int addhip_actor_head_slabs(int32_t rows);
int addhip_actor_head(const addhip_actor_head_t* p, void* stream);
/* count of rand_mask == 1 -> out[0] */
int addhip_count_mask(const float* rand_mask, int32_t M, float* out, void* stream);
/* PPOAgent._compute_critic_loss (ppo_agent.py:209-219): v = H.w+b; dv = scale*2(v-tar)/M;
 * dZ[m,:] = dv*w*(H>0) ; stats[0] += sum (tar-v)^2 ; dv_out for the dw/db reductions */
/* (dZ may be NULL: addhip_head_backward then produces it together with the head's gradients) */
int addhip_critic_head(const float* H, int32_t ld, int32_t K, int32_t M, const float* w, const float* b,
                       const float* tar, float loss_scale, float* dZ, float* dv_out, float* stats, void* stream);
/* ADDAgent._compute_disc_loss head part (add_agent.py:141-202, amp_agent.py:177-192): logits, BCE(0.1) on the
 * M agent/demo differences and BCE(0.9) on the single zero-difference row `h_pos`; writes dlogit[M], dlogit_pos[1];
 * stats += {sum bce_neg, bce_pos, sum logit_neg, logit_pos, #neg<0, pos>0} */
int addhip_disc_head(const float* H, int32_t ld, int32_t K, int32_t M, const float* h_pos, const float* w,
                     const float* b, float loss_scale, float* dlogit, float* dlogit_pos, float* stats, void* stream);
/* Backward of a scalar head (critic value / discriminator logit) in one pass over the last hidden layer H [rows,ld]:
 * dZ[r,k] = (H[r,k] > 0) ? v[r]*w[k] : 0 (the layer's pre-activation gradient; what addhip_outer_mask writes), and the
 * column sums that autograd would produce for the head weight (sum_r v[r] H[r,k]), the head bias (sum_r v[r]) and the
 * layer's bias (sum_r dZ[r,k]), ACCUMULATED by atomics into dW_head[K], db_head[1], db_top[K] (caller zeroes; any output
 * may be NULL).  Replaces outer_mask + weighted_col_sum + 2 x col_sum and three of their four passes over H.  K <= 1024.
 * dZ16 (optional): the same dZ rounded to bf16 (nearest even), what the bf16-storage backward GEMMs read. */
int addhip_head_backward(const float* v, const float* w, const float* H, int32_t ld, int32_t K, int64_t rows,
                         float* dZ, uint16_t* dZ16, int32_t planes16 /* ADDHIP_STORE_* format of dZ16 */, float* dW_head, float* db_head, float* db_top,
                         uint32_t* amax /* optional: max |dZ| into ADDHIP_AMAX_SLOTS slots (caller zeroes) */,
                         float* ordered_scratch /* optional: >= ADDHIP_HEAD_BWD_BLOCKS * (2 K + 4) floats; the three sums are then formed in a
                         fixed order (per-workgroup partials, added workgroup by workgroup by a second launch) instead of by float atomics:
                         bit-identical from run to run */, void* stream);
#define ADDHIP_HEAD_BWD_BLOCKS 256

/* out[m,k] = v[m] * w[k] * (H[m,k] > 0)    (back through a 1-wide head into the last hidden layer) */
int addhip_outer_mask(const float* v, const float* w, const float* H, int32_t ld, int32_t K, int64_t rows,
                      float* out, void* stream);
/* out[m,k] = w[k] * (H[m,k] > 0)          (a2 of the gradient-penalty chain); out and/or its bf16 copy out16 */
int addhip_bcast_mask(const float* w, const float* H, int32_t ld, int32_t K, int64_t rows, float* out, uint16_t* out16, int32_t planes16, uint32_t* amax /* as above */,
                      void* stream);
/* gradient penalty (add_agent.py:166-178): n=sqrt(|g|^2+1e-8); G = coef*2(n-1)/n * g / M (fp32 G and/or its bf16 copy G16); stats[0] += sum (n-1)^2 */
int addhip_grad_penalty(const float* g, int32_t ld, int32_t dim, int32_t M, float coef, float* G, uint16_t* G16, int32_t planes16, float* stats,
                        uint32_t* amax /* as above: max |G| */, void* stream);
/* out[k] (+)= scale * sum_m v[m]*(mask? (Hmask[m,k]>0):1)*X[m,k]   (dw of 1-wide heads) */
int addhip_weighted_col_sum(const float* v, const float* X, int32_t ld, int32_t K, int64_t rows, float* out,
                            float scale, int32_t accumulate, void* stream);
/* grad[i] += coef * w[i]  (2*lambda*W terms: disc_logit_reg, disc_weight_decay) ; stats_out += sum w^2 (optional) */
int addhip_l2_grad(const float* w, float* grad, int64_t count, float coef, float* sumsq_out, void* stream);

/* MPOptimizer._clip_grads (mp_optimizer.py:45-46) = torch.nn.utils.clip_grad_norm_ over the whole flat gradient:
 * grad *= min(1, max_norm / (||grad||_2 + 1e-6)).  scratch: >= 2 floats, 8-byte aligned (device); norm_out[0] (or NULL)
 * receives the unclipped norm. */
int addhip_grad_clip(float* grad, int64_t count, float max_norm, float* scratch, float* norm_out, void* stream);

/* torch.optim.AdamW step on a flat buffer (mp_optimizer.py:14-40; lr, betas (0.9,0.999), eps 1e-8) */
int addhip_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* torch.optim.SGD(momentum=0.9) step on a flat buffer (optimizer.type "SGD", mp_optimizer.py:33-36): g += wd*p; buf = (step==1) ? g : momentum*buf + g;
 * p -= lr*buf */
int addhip_sgd(float* param, const float* grad, float* momentum_buf, int64_t count, float lr, float momentum, float weight_decay,
               int32_t step, void* stream);

/* MPOptimizer.step (mp_optimizer.py:14-46: zero_grad ... step) as one launch over the flat buffers: the AdamW or SGD(momentum) update of
 * addhip_adamw / addhip_sgd element for element, and in the same pass (both optional)
 *   param16   the bf16 shadow of the updated parameters, round to nearest even (bf16-storage mode: what the next forward GEMMs read);
 *   zero_grad grad[i] = 0 after it was read: the zero_grad with which the NEXT step begins.
 * state1 = exp_avg (AdamW) / momentum buffer (SGD), state2 = exp_avg_sq (AdamW; unused for SGD); beta1 doubles as the SGD momentum. */
enum { ADDHIP_OPT_ADAMW = 0, ADDHIP_OPT_SGD = 1 };
typedef struct {
  int32_t type;
  float* param; float* grad; float* state1; float* state2;
  int64_t count;
  float lr, beta1, beta2, eps, weight_decay;
  int32_t step;            /* 1-based */
  uint16_t* param16;       /* or NULL */
  int32_t zero_grad;
  int32_t param16_planes;  /* ADDHIP_STORE_* format of param16 (0 = ADDHIP_STORE_BF16); X3: 3 * count uint16, count % 8 == 0 */
} addhip_optimizer_t;
int addhip_optimizer_step(const addhip_optimizer_t* o, void* stream);

/* ReturnTracker running means over the T steps of an iteration (base_agent.py:596-621):
 * ep_stats [T,3] -> state {episodes, mean_return, mean_ep_len} updated step by step */
int addhip_return_tracker_fold(const float* ep_stats, int32_t T, float* state, void* stream);

/* ---- recorded plans and multi-stream schedules: the host-side runtime of the update step.
 *
 * The reference runs its optimiser step call by call from Python (PPOAgent._update_model / _compute_loss, learning/ppo_agent.py:171-275;
 * ADDAgent._compute_disc_loss, learning/add/add_agent.py:141-202).  Here a host records the step ONCE and replays it with one call per
 * stream section, in any language:
 *
 *   addhip_plan_create(&p); addhip_plan_record_begin(p);
 *   ... any stream-taking entry point of this header, called on this thread: its arguments are checked as usual, then the call is
 *       APPENDED to the plan instead of launched (the stream argument is ignored).  Scalars and device addresses are kept by value;
 *       every parameter block (addhip_gemm_t, addhip_motion_t, ...) and host-side table is COPIED at record time, so the caller's
 *       structs need not outlive the call.  The composite entry points below (addhip_mlp_forward, ...) record their launches the same way ...
 *   addhip_plan_record_end(p);
 *   addhip_plan_run(p, first, last, stream);      replays calls [first, last) on `stream` (last < 0: to the end)
 *
 * A plan holds host memory only (no device allocation, no HIP object); replaying it is exactly the recorded launches, so a replay
 * is hipGraph-capturable like the calls themselves.  Plans are not thread-safe: record and run a plan from one thread at a time. ---- */
typedef struct addhip_plan addhip_plan_t;
int addhip_plan_create(addhip_plan_t** out);
/* (refused, plan left alive, while a schedule created over the plan has not been destroyed: destroy schedules first) */
int addhip_plan_destroy(addhip_plan_t* plan);
int addhip_plan_record_begin(addhip_plan_t* plan);
int addhip_plan_record_end(addhip_plan_t* plan);
/* number of recorded calls (>= 0) */
int addhip_plan_size(const addhip_plan_t* plan);
/* introspection (roofline accounting, tools): the entry point's name; for a GEMM launch its problem descriptor(s) -> out[0..n), returns
 * n (0: not a GEMM; 2-4: a grouped launch) */
const char* addhip_plan_call_name(const addhip_plan_t* plan, int32_t index);
int addhip_plan_call_gemms(const addhip_plan_t* plan, int32_t index, addhip_gemm_t* out, int32_t capacity);
int addhip_plan_run(const addhip_plan_t* plan, int32_t first, int32_t last, void* stream);

/* A schedule runs ranges of a plan on several streams with the event dependencies between them -- one optimiser step of the ADD agent is
 * eight such sections on four streams -- and tells the host where in the issue order a gradient bucket is final, so that
 * the host issues its exchange step (RCCL all-reduce) there, on that stream, overlapped with the sections that follow.
 * Sections are issued in list order:  wait for section `wait_before` (-1: none) -> calls [first, last) on streams[stream] -> wait for
 * section `wait_after` (-1: none) -> on_bucket(user, bucket, streams[stream]) if bucket >= 0.  Before the first section every stream
 * is made to wait for what streams[0] holds; after the last, streams[0] waits for all the others.  Only hipEventRecord /
 * hipStreamWaitEvent are used (events created once, by addhip_schedule_create): capturable. */
#define ADDHIP_MAX_STREAMS 8
typedef struct {
  int32_t stream;            /* index into the streams[] of addhip_schedule_run */
  int32_t first, last;       /* plan calls [first, last); an empty range is allowed (a bucket that is final without further calls) */
  int32_t wait_before;       /* index of an EARLIER section this one starts after, or -1 */
  int32_t wait_after;        /* index of an earlier section whose end this one's bucket / successors also depend on, or -1 */
  int32_t bucket;            /* >= 0: passed to on_bucket when the section has been issued; -1: none */
} addhip_section_t;
typedef struct addhip_schedule addhip_schedule_t;
typedef void (*addhip_bucket_fn)(void* user, int32_t bucket, void* stream);
int addhip_schedule_create(const addhip_plan_t* plan, const addhip_section_t* sections, int32_t count, int32_t num_streams, addhip_schedule_t** out);
int addhip_schedule_destroy(addhip_schedule_t* schedule);
int addhip_schedule_run(addhip_schedule_t* schedule, void* const* streams, addhip_bucket_fn on_bucket, void* user);

/* ---- composite entry points: the networks' passes and the two loss sections of an optimiser step, assembled by the library from the
 *      entry points above (no kernels of their own), so that a host in any language runs the measured step without re-deriving its ~70
 *      launches.  Called with a stream they launch; called between addhip_plan_record_begin / _end they record their launches into the plan.
 *      Restates PPOModel / ADDModel's Sequential stacks (learning/ppo_model.py:13-59, learning/add/add_model.py:12-46) and the loss /
 *      backward of PPOAgent._compute_actor_loss / _compute_critic_loss (learning/ppo_agent.py:194-275, base_agent.py:522-546) and
 *      ADDAgent._compute_disc_loss (learning/add/add_agent.py:141-202, amp_agent.py:177-192). ---- */
#define ADDHIP_MLP_MAX_HIDDEN 4
typedef struct {
  int32_t num_hidden;                     /* Linear+ReLU layers, 1..ADDHIP_MLP_MAX_HIDDEN */
  int32_t in_dim, in_ld;                  /* input columns / row stride of the input rows (pad columns hold zeros) */
  int32_t hidden[ADDHIP_MLP_MAX_HIDDEN];  /* layer widths */
  int32_t head_rows;                      /* rows of the head weight: 32 for the 29-wide action-mean head (zero rows as padding), 1 for a scalar head */
  int32_t precision;                      /* ADDHIP_PREC_* of this net's GEMMs */
  int32_t storage;                        /* 0: fp32 operands (precision ADDHIP_PREC_BF16 then means one bf16 product per term on operands cut
                                             to bf16 on the way into LDS); ADDHIP_STORE_BF16 (with precision ADDHIP_PREC_BF16): bf16 STORAGE,
                                             the *16 buffers below; ADDHIP_STORE_BF16X3: they hold plane storage (3 uint16 per value) and every
                                             hidden GEMM runs on them (csrc/gemm_x3.hip); precision is then ADDHIP_PREC_BF16X3 */
  /* parameters: W[i] [hidden[i], in_ld | hidden[i-1]] row major, b[i] [hidden[i]], Wh [head_rows, hidden[last]], bh; g* = their gradients */
  const float* W[ADDHIP_MLP_MAX_HIDDEN]; const float* b[ADDHIP_MLP_MAX_HIDDEN]; const float* Wh; const float* bh;
  float* gW[ADDHIP_MLP_MAX_HIDDEN]; float* gb[ADDHIP_MLP_MAX_HIDDEN]; float* gWh; float* gbh;
  /* bf16 storage: W16[i] = bf16(W[i]), same layout; W16t[i] = its transpose [in, out] for every i a dX GEMM or the penalty chain reads */
  const uint16_t* W16[ADDHIP_MLP_MAX_HIDDEN]; const uint16_t* W16t[ADDHIP_MLP_MAX_HIDDEN];
  /* caller-allocated workspace for up to rows_cap rows */
  int32_t rows_cap;
  float* h[ADDHIP_MLP_MAX_HIDDEN];        /* [rows_cap, hidden[i]] activations (bf16 storage: only the last layer's is written) */
  float* dz[ADDHIP_MLP_MAX_HIDDEN];       /* [rows_cap, hidden[i]] pre-activation gradients (fp32-operand modes) */
  uint32_t* hbits[ADDHIP_MLP_MAX_HIDDEN]; /* [rows_cap, ceil(hidden[i]/32)] ReLU sign bits (forward with sign_bits -> backward) */
  uint16_t* h16[ADDHIP_MLP_MAX_HIDDEN]; uint16_t* dz16[ADDHIP_MLP_MAX_HIDDEN];  /* bf16 storage */
  float* slabs; int64_t slab_floats;      /* split-K scratch of the weight gradients: 2 * split * out * in floats of the largest layer */
  float* slabs_top;                       /* optional second scratch: the top layer's weight gradient may then run beside the rest (or NULL) */
  /* optional: bias_replica_rows (16) x max(hidden) floats, ZERO at the first use (the library leaves them zero): the dX GEMMs add their
   * bias-gradient column sums there, spread over the rows, and the split-K combine that follows sums them into gb (NULL: plain atomics on gb) */
  float* bias_replicas; int32_t bias_replica_rows;
  /* bf16 storage, optional: this net's transposed shadows are rewritten (addhip_shadow_refresh, flat16 = NULL) between its forward and
   * backward pass, on its own stream.  t_offset / t_rows / t_cols: HOST arrays of t_count entries (copied when recorded). */
  const float* flat_params; uint16_t* flat_trans16; int64_t flat_count;
  const int64_t* t_offset; const int32_t* t_rows; const int32_t* t_cols; int32_t t_count;
  /* precision = ADDHIP_PREC_F16X2: the tracked operand maxima its GEMMs scale by.  amax: workspace of ADDHIP_MLP_AMAX_TENSORS x
   * ADDHIP_AMAX_SLOTS uint32 (tensor t at amax + t * ADDHIP_AMAX_SLOTS: h[i] at ADDHIP_MLP_AMAX_H + i, dz[i] at ADDHIP_MLP_AMAX_DZ + i, the
   * penalty chain's a2 / a1 / G / e1 behind them), zeroed by every forward pass before its GEMMs rewrite it; w_amax: ADDHIP_AMAX_SLOTS
   * slots bounding every weight the net reads (e.g. addhip_amax_f32 over the flat parameter buffer after each optimiser step).  Either
   * NULL: the net's GEMMs run as ADDHIP_PREC_BF16X3. */
  uint32_t* amax; const uint32_t* w_amax;
  /* deterministic != 0: every reduction behind this net's gradients runs in a fixed order (agent.deterministic) -- the reference's CPU
   * path is reproducible under a seed (learning/mp_optimizer.py:14-23 on torch-CPU).  Needs bias_replica_rows >= ceil(rows_cap / 32) (each
   * 32-row block of a dX GEMM then owns its replica row: one add per slot, summed in row order by the combine) and ordered_scratch:
   * >= ADDHIP_HEAD_BWD_BLOCKS * (2 * max(hidden) + 4) floats (addhip_head_backward, addhip_col_sum_ordered).  Loss diagnostics (stats) are
   * still accumulated by atomics: logged scalars may differ in their last bits, gradients and parameters do not. */
  int32_t deterministic; float* ordered_scratch;
} addhip_mlp_t;
enum { ADDHIP_MLP_AMAX_H = 0, ADDHIP_MLP_AMAX_DZ = 4,
       ADDHIP_MLP_AMAX_A = 8 /* + layer: the penalty chain's a[i] */, ADDHIP_MLP_AMAX_G = 12, ADDHIP_MLP_AMAX_E = 13 /* + layer: e[i], i < last */,
       ADDHIP_MLP_AMAX_TENSORS = 16 };

/* h[last] = MLP(x) for `rows` rows (<= rows_cap): one GEMM per layer with fused bias + ReLU (+ fused (x - a_mean) / a_std on the first
 * layer's input when a_mean is given: Normalizer.normalize, normalizer.py:107-110; not with bf16 storage, whose input x16 is the
 * already normalised rows rounded to bf16).  sign_bits: also write hbits (a backward pass follows). */
int addhip_mlp_forward(const addhip_mlp_t* net, const float* x, const uint16_t* x16, int64_t rows, const float* a_mean, const float* a_std,
                       int32_t sign_bits, const uint32_t* x_amax /* ADDHIP_PREC_F16X2: bound of max |x| (ADDHIP_AMAX_SLOTS slots) or NULL */, void* stream);

/* Backward pass through the hidden stack.  In: dz[last] (bf16 storage: dz16[last]) = d loss / d pre-activation of the last hidden layer,
 * written by the caller (the loss sections below do).  Out: gW[i], gb[i] for every layer.  Weight gradients are split-K GEMMs combined
 * by addhip_slab_reduce; a layer's bias gradient is the fused column sum of the dX GEMM above it. */
enum {
  ADDHIP_BWD_GRADS_ZEROED = 1,   /* the caller zeroed the whole gradient buffer at the start of the step (else each gb is cleared here) */
  ADDHIP_BWD_TOP_BIAS_DONE = 2,  /* the producer of dz[last] also accumulated gb[last] */
  ADDHIP_BWD_ACCUMULATE_DW = 4,  /* weight gradients are ADDED to what gW holds (L2 terms written there earlier in the step) */
  ADDHIP_BWD_TOP_CAST_DONE = 8,  /* bf16 storage: dz16[last] is already written (else it is rounded from dz[last] first) */
  ADDHIP_BWD_SIGN_BITS = 16,     /* the forward pass of this step wrote hbits: masks are read from them */
  ADDHIP_BWD_TOP_BIAS_REPLICAS = 32 /* the producer of dz[last] left gb[last] as column sums spread over bias_replicas: the top combine folds them in */
};
typedef struct {                 /* a second product accumulated into a layer's weight gradient: gW += A^T B over `rows` rows */
  const void* A; int32_t lda; const void* B; int32_t ldb; int64_t rows;   /* fp32, or bf16 with bf16 storage; A == NULL: none */
  const uint32_t* a_amax; const uint32_t* b_amax;                         /* ADDHIP_PREC_F16X2: their tracked maxima (or NULL) */
} addhip_extra_dw_t;
typedef struct {                 /* where things are final, as launch counts from this call's first launch (for schedules) */
  int32_t launches;              /* launches issued by the call */
  int32_t early;                 /* after this many, every gradient of the net but W[0] / b[0] is final (0 if the net has one layer) */
  int32_t dw_first[ADDHIP_MLP_MAX_HIDDEN], dw_last[ADDHIP_MLP_MAX_HIDDEN];  /* launches [first, last) form layer i's weight gradient */
} addhip_mlp_marks_t;
int addhip_mlp_backward(const addhip_mlp_t* net, const float* x, const uint16_t* x16, int64_t rows, const addhip_extra_dw_t* extra /* [num_hidden] or NULL */,
                        int32_t flags, addhip_mlp_marks_t* marks /* or NULL */, const uint32_t* x_amax /* as for addhip_mlp_forward */, void* stream);

/* The actor's and the critic's sections of one optimiser step on a gathered minibatch (addhip_gather_minibatch's outputs): forward,
 * loss heads (addhip_actor_loss; addhip_critic_head), backward.  Needs the flat gradient zeroed (addhip_fill_zero / the zero_grad of
 * addhip_optimizer_step).  grad_scale = 1 / world size: with it the SUM of the ranks' gradients is their mean. */
typedef struct {
  const addhip_mlp_t* actor; const addhip_mlp_t* critic;
  int32_t rows;                                   /* minibatch rows Mb */
  const float* norm_obs; const uint16_t* norm_obs16;   /* [Mb, in_ld] normalised observations (+ bf16 copy with bf16 storage) */
  const uint32_t* norm_obs_amax;                  /* ADDHIP_PREC_F16X2: their tracked maximum (addhip_gather_t.obs_amax) or NULL */
  const float* norm_action;                       /* [Mb, 32] */
  const float* old_logp; const float* adv; const float* tar_val; const float* rand_mask;   /* [Mb] */
  float action_std, logp_const, ppo_clip_ratio, action_bound_weight, action_reg_weight, critic_loss_weight, grad_scale;
  const float* dist; float* g_logstd;             /* actor_std_type CONSTANT: addhip_dist_refresh's vector and the log-std's gradient [32] (NULL: FIXED) */
  float action_entropy_weight;                    /* actor_std_type VARIABLE (actor->head_rows == 64): the entropy bonus's weight (other types: its
                                                     gradient is zero / added by the host behind the exchange) */
  int32_t head_precision;                         /* ADDHIP_PREC_* of the three 32-wide head GEMMs (fp32 operands in every mode) */
  float* mean; float* d_mean;                     /* workspace [Mb, actor->head_rows] each (32; 64 with a log-std head: mean | log-std) */
  float* dv;                                      /* workspace [Mb] */
  float* num_valid;                               /* workspace [1] */
  float* stats;                                   /* [32] loss diagnostics, accumulated over the steps of an iteration (slots: learn.hip) */
} addhip_ppo_loss_t;
typedef struct {
  int32_t launches;                               /* actor's section = launches [0, actor_end), critic's = [actor_end, launches) */
  int32_t actor_end, actor_early, critic_early;   /* *_early: that net's gradient but its first layer is final (absolute launch counts of this call) */
} addhip_ppo_marks_t;
int addhip_ppo_loss_fwd_bwd(const addhip_ppo_loss_t* d, addhip_ppo_marks_t* marks /* or NULL */, void* stream);

/* The discriminator's section: L2 terms (logit regularisation, weight decay) into the zeroed gradient, forward over Mb agent/demo
 * differences + one zero-difference row, logit loss + head backward, the gradient-penalty chain with its second-order terms
 * (hand-derived double backward of add_agent.py:166-178, any number of hidden layers n <= ADDHIP_MLP_MAX_HIDDEN:
 *   a[n-1] = w_head * m[n-1];  a[i-1] = (a[i] W[i]) * m[i-1];  g = a[0] W[0] = d logit / d input;  penalty = mean (|g| - 1)^2;
 *   G = d penalty / d g;  e[0] = (G W[0]^T) * m[0];  e[i] = (e[i-1] W[i]^T) * m[i];  dW[0] += a[0]^T G, dW[i] += a[i]^T e[i-1], d w_head += sum_rows e[n-1]
 * with m[i] the ReLU masks of the forward pass), backward. */
typedef struct {
  const addhip_mlp_t* disc;
  int32_t rows;                                   /* Mb; the net runs Mb + 1 rows (rows_cap >= Mb + 1) */
  int32_t disc_dim;                               /* columns of a difference row that carry data (<= disc->in_ld) */
  const float* norm_diff; const uint16_t* norm_diff16;   /* [Mb + 1, in_ld]; row Mb is all zeros */
  const uint32_t* norm_diff_amax;                 /* ADDHIP_PREC_F16X2: their tracked maximum (addhip_gather_t.diff_amax) or NULL */
  float loss_scale;                               /* disc_loss_weight * grad_scale */
  float logit_reg, grad_penalty, weight_decay;
  float* dlogit;                                  /* workspace [Mb + 1] */
  float* a[ADDHIP_MLP_MAX_HIDDEN]; float* e[ADDHIP_MLP_MAX_HIDDEN];  /* workspace, per hidden layer i: a[i], e[i] [Mb, hidden[i]] (e[last] always fp32) */
  float* g; float* G;                                                   /* workspace [Mb, in_ld] each */
  uint16_t* a16[ADDHIP_MLP_MAX_HIDDEN]; uint16_t* e16[ADDHIP_MLP_MAX_HIDDEN]; uint16_t* G16;   /* bf16 / plane storage: a[i], e[i < last], G as 16-bit rows
                                                                           instead (their fp32 pointers may then be NULL) */
  float* stats;
} addhip_disc_loss_t;
typedef struct {
  int32_t launches;
  int32_t head, chain, backward, top_dw_first, top_dw_last;   /* launch counts at which: the logit loss starts | the penalty chain starts | the
                                                                  backward pass starts | the top layer's weight-gradient group is [first, last) */
} addhip_disc_marks_t;
int addhip_disc_loss_fwd_bwd(const addhip_disc_loss_t* d, addhip_disc_marks_t* marks /* or NULL */, void* stream);

/* The schedule of one optimiser step recorded as  addhip_ppo_loss_fwd_bwd  then  addhip_disc_loss_fwd_bwd  into a plan whose size was
 * `base` before the first: ten sections on four streams (actor | critic | discriminator x 2), buckets 0 = actor but its first layer,
 * 1 = critic likewise, 3 = the two first layers, 2 = discriminator -- reported in that order (0, 1, 3, 2: the order in which they become
 * final, so that only the last one's exchange has no launches left to overlap).  -> out[0..10), returns 10. */
int addhip_update_schedule(int32_t base, const addhip_ppo_marks_t* ppo, const addhip_disc_marks_t* disc, addhip_section_t* out, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif

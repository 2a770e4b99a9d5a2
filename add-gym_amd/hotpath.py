"""Struct builders for the C ABI: turn add-gym's config keys into the parameter blocks of
include/addhip.h.  Pure host logic (no GPU needed)."""
import numpy as np

from . import _lib as L

F = np.float32


VEL_W = 35  # root_vel 3 + root_ang_vel 3 + dof_vel 29


def obs_dims(task):
    """(obs_dim, disc_dim) (add_observation.py:231-274, 422-459, 520-575)."""
    hc = 1 if task.get("root_height_obs", False) else 0
    k = len(task.get("tar_obs_steps", [1])) if task.get("enable_tar_obs", False) else 0
    vel = VEL_W if task.get("enable_vel_obs", False) else 0
    phase = (1 + 2 * int(task.get("num_phase_encoding", 0))) if task.get("enable_phase_obs", True) else 0
    obs_dim = hc + 6 + L.NUM_DOF + vel + phase + k * ((3 if hc else 2) + 6 + L.NUM_DOF)
    return obs_dim, disc_steps(task) * (L.DISC_STEP_W + vel)


def disc_steps(task):
    """task.num_disc_obs_steps: poses per discriminator observation = depth of the history ring (add_observation.py:276-294)."""
    return int(task.get("num_disc_obs_steps", L.HIST))


def pad4(n):
    return (n + 3) // 4 * 4


def pad16(n):
    """Row strides of obs / disc-obs buffers: whole 64-byte groups, so that the K extent of the first-layer GEMMs is a whole
    number of 16-deep MFMA stages (no tail stage) and rows start on 64-byte boundaries."""
    return (n + 15) // 16 * 16


def check_supported(task):
    """The HIP path covers the observation flags of configs/task/pose.yaml plus the global/height toggles.
    Anything else fails loudly instead of silently computing something different."""
    if task.get("enable_phase_obs", True) and int(task.get("num_phase_encoding", 0)) > 8:
        raise NotImplementedError("task.num_phase_encoding must be <= 8")
    if not 2 <= disc_steps(task) <= L.HIST_MAX:
        raise NotImplementedError(f"task.num_disc_obs_steps must be in 2..{L.HIST_MAX} (the env-step kernel stages at most {L.HIST_MAX - 1} history rows)")
    if task.get("visualize_ref_char", False):
        raise NotImplementedError("task.visualize_ref_char needs a viewer (out of scope)")
    if len(task.get("tar_obs_steps", [1])) > L.MAX_TAR:
        raise NotImplementedError(f"at most {L.MAX_TAR} target steps")
    jw = task.get("joint_err_w", None)
    if jw is not None and len(jw) != L.NUM_DOF:
        raise NotImplementedError("task.joint_err_w must list one weight per joint (29 one-dof joints for G1)")


def make_task(task, dt, max_episode_length=None):
    check_supported(task)
    t = L.TaskT()
    t.dt = dt
    t.global_obs = int(bool(task.get("global_obs", False)))
    t.root_height_obs = int(bool(task.get("root_height_obs", False)))
    steps = list(task.get("tar_obs_steps", [1])) if task.get("enable_tar_obs", False) else []
    t.num_tar_steps = len(steps)
    # fp32 products exactly as the reference forms them (add_observation.py:215, 366-369)
    tar = (F(dt) * np.asarray(steps, F)).astype(F)
    for i, v in enumerate(tar):
        t.tar_dt[i] = float(v)
    t.num_disc_obs_steps = disc_steps(task)
    demo = (F(-dt) * np.arange(t.num_disc_obs_steps, dtype=F))[::-1].astype(F)
    for i, v in enumerate(demo):
        t.demo_dt[i] = float(v)
    t.max_episode_length = float(task.get("max_episode_length", max_episode_length if max_episode_length is not None else 0.0))
    t.enable_early_termination = int(bool(task["enable_early_termination"]))
    t.pose_termination = int(bool(task.get("pose_termination", False)))
    t.pose_termination_dist = float(task.get("pose_termination_dist", 1.0))
    t.pose_w, t.vel_w = float(task["reward_pose_w"]), float(task["reward_vel_w"])
    t.root_pose_w, t.root_vel_w = float(task["reward_root_pose_w"]), float(task["reward_root_vel_w"])
    t.pose_scale, t.vel_scale = float(task["reward_pose_scale"]), float(task["reward_vel_scale"])
    t.root_pose_scale, t.root_vel_scale = float(task["reward_root_pose_scale"]), float(task["reward_root_vel_scale"])
    t.enable_vel_obs = int(bool(task.get("enable_vel_obs", False)))
    t.enable_phase_obs = int(bool(task.get("enable_phase_obs", True)))
    t.num_phase_encoding = int(task.get("num_phase_encoding", 0))
    t.obs_dim, t.disc_dim = obs_dims(task)
    t.obs_stride, t.disc_stride = pad16(t.obs_dim), pad16(t.disc_dim)
    return t


# agent.matmul_precision -> ADDHIP_PREC_* (how addhip_gemm_f32 forms its products; a property of each descriptor)
# ("bf16x3": every GEMM splits its fp32 operands on the fly, csrc/gemm_split.hip; "bf16x3_planes": the update step runs on plane-stored
# operands, csrc/gemm_x3.hip -- same products and error, measured slower end to end: DESIGN.md section 4)
PRECISIONS = {"fp32": L.PREC_F32, "f32": L.PREC_F32, "bf16x3": L.PREC_BF16X3, "bf16x3_planes": L.PREC_BF16X3, "bf16x2": L.PREC_BF16X2, "bf16": L.PREC_BF16,
              "f16x2": L.PREC_F16X2}  # f16x2: two-way fp16 split on per-tensor power-of-two scales, four products (fp32-class: 2^-21 |a||b| per product)


def gemm(M, N, K, A, lda, a_kc, B, ldb, b_kc, C, ldc, epilogue=L.EPI_NONE, bias=None, mask=None, ldmask=0, a_mean=None, a_std=None,
         split_k=1, alpha=1.0, colsum=None, precision=L.PREC_F32, relu_bits=None, mask_bits=None, ldbits=0, accumulate=0, operands_bf16=0, C16=None, ldc16=0, hint=0,
         colsum_replicas=0, ldcs=0, c16_planes=0, a_amax=None, b_amax=None, amax_out=None):
    """Descriptor for addhip_gemm_f32: C[M,N] = alpha * sum_k A(m,k) B(n,k).  Pointers are raw addresses."""
    return L.GemmT(M, N, K, A, lda, int(a_kc), B, ldb, int(b_kc), C, ldc, epilogue, bias, mask, ldmask, a_mean, a_std, split_k, alpha, colsum,
                   precision, relu_bits, mask_bits, ldbits, int(accumulate), int(operands_bf16), C16, ldc16, int(hint), int(colsum_replicas), int(ldcs), int(c16_planes), a_amax, b_amax, amax_out)

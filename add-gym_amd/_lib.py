"""ctypes binding of libaddhip.so (include/addhip.h).  The product path has no CPU fallback:
if the HIP library is missing or a call fails, this raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaddhip.so")

MAX_TAR = 8
HIST = 3        # default task.num_disc_obs_steps
HIST_MAX = 4    # ADDHIP_HIST_MAX
DONE_NULL, DONE_FAIL, DONE_SUCC, DONE_TIME = 0, 1, 2, 3  # base_agent.py:16-20
POSE_W = 36
NUM_DOF = 29
DISC_STEP_W = 9 + NUM_DOF  # pos3 + tan/norm 6 + dof 29


class AddhipError(RuntimeError):
    pass


f32p = C.c_void_p  # all device pointers travel as integers (tensor.data_ptr())


class MotionT(C.Structure):
    _fields_ = [("pose", f32p), ("vel", f32p), ("clip_start", f32p), ("clip_steps", f32p), ("clip_len", f32p), ("clip_loop", f32p),
                ("num_clips", C.c_int32), ("total_steps", C.c_int32), ("reference_compat", C.c_int32), ("dt_inv", C.c_float)]


class TaskT(C.Structure):
    _fields_ = [("dt", C.c_float), ("global_obs", C.c_int32), ("root_height_obs", C.c_int32), ("num_tar_steps", C.c_int32),
                ("tar_dt", C.c_float * MAX_TAR), ("demo_dt", C.c_float * HIST_MAX), ("max_episode_length", C.c_float),
                ("enable_early_termination", C.c_int32), ("pose_termination", C.c_int32), ("pose_termination_dist", C.c_float),
                ("pose_w", C.c_float), ("vel_w", C.c_float), ("root_pose_w", C.c_float), ("root_vel_w", C.c_float),
                ("pose_scale", C.c_float), ("vel_scale", C.c_float), ("root_pose_scale", C.c_float), ("root_vel_scale", C.c_float),
                ("obs_dim", C.c_int32), ("obs_stride", C.c_int32), ("disc_dim", C.c_int32), ("disc_stride", C.c_int32),
                ("enable_vel_obs", C.c_int32), ("enable_phase_obs", C.c_int32), ("num_phase_encoding", C.c_int32), ("num_disc_obs_steps", C.c_int32)]


class EnvT(C.Structure):
    _fields_ = [("num_envs", C.c_int32), ("sim_pose", f32p), ("sim_vel", f32p), ("time", f32p), ("time_off", f32p), ("motion_id", f32p),
                ("hist", f32p), ("hist_vel", f32p), ("done", f32p), ("contact", f32p), ("ref_pose", f32p), ("ref_vel", f32p), ("ret_acc", f32p), ("len_acc", f32p), ("dof_err_w", f32p)]


class StepOutT(C.Structure):
    _fields_ = [("obs", f32p), ("obs_next_in", f32p), ("obs_timeout", f32p), ("disc_obs", f32p), ("disc_demo", f32p), ("reward", f32p), ("done", f32p),
                ("motion_id_rec", f32p), ("motion_time_rec", f32p), ("ep_stats", f32p)]


class SamplerT(C.Structure):
    _fields_ = [("errors", f32p), ("seg_size", f32p), ("clip_cdf", f32p), ("num_segments", C.c_int32), ("temperature", C.c_float),
                ("min_start_time", C.c_float), ("rand_reset", C.c_int32), ("temp_bits", f32p), ("err_sum", f32p), ("err_cnt", f32p)]


class GemmT(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("A", f32p), ("lda", C.c_int32), ("a_kcontig", C.c_int32),
                ("B", f32p), ("ldb", C.c_int32), ("b_kcontig", C.c_int32), ("C", f32p), ("ldc", C.c_int32), ("epilogue", C.c_int32),
                ("bias", f32p), ("mask", f32p), ("ldmask", C.c_int32), ("a_mean", f32p), ("a_std", f32p), ("split_k", C.c_int32),
                ("alpha", C.c_float), ("colsum", f32p), ("precision", C.c_int32),
                ("relu_bits", f32p), ("mask_bits", f32p), ("ldbits", C.c_int32), ("accumulate", C.c_int32),
                ("operands_bf16", C.c_int32), ("C16", f32p), ("ldc16", C.c_int32), ("hint", C.c_int32), ("colsum_replicas", C.c_int32), ("ldcs", C.c_int32),
                ("c16_planes", C.c_int32), ("a_amax", f32p), ("b_amax", f32p), ("amax_out", f32p)]


class GatherT(C.Structure):
    _fields_ = [("idx", f32p), ("count", C.c_int32), ("obs", f32p), ("obs_stride", C.c_int32), ("obs_dim", C.c_int32), ("obs_mean", f32p),
                ("obs_std", f32p), ("action", f32p), ("a_mean", f32p), ("a_std", f32p), ("a_logp", f32p), ("adv", f32p), ("tar_val", f32p),
                ("rand_mask", f32p), ("disc_obs", f32p), ("disc_demo", f32p), ("disc_stride", C.c_int32), ("disc_dim", C.c_int32),
                ("mean_abs", f32p), ("min_diff", C.c_float), ("norm_obs", f32p), ("norm_action", f32p), ("o_logp", f32p), ("o_adv", f32p),
                ("o_tar_val", f32p), ("o_mask", f32p), ("norm_diff", f32p), ("norm_obs16", f32p), ("norm_diff16", f32p), ("planes16", C.c_int32),
                ("obs_amax", f32p), ("diff_amax", f32p)]


class RigidModelT(C.Structure):
    _fields_ = [("num_bodies", C.c_int32), ("num_points", C.c_int32), ("body", f32p), ("topo", f32p), ("points", f32p), ("dt", C.c_float),
                ("substeps", C.c_int32), ("gravity", C.c_float), ("contact_stiffness", C.c_float), ("contact_damping", C.c_float),
                ("friction", C.c_float), ("friction_vel_eps", C.c_float), ("limit_stiffness", C.c_float), ("max_torque", C.c_float),
                ("limit_margin", C.c_float), ("termination_mask", C.c_uint32), ("env_scale", f32p), ("chains", f32p)]


class RigidDrT(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("resample_interval", C.c_int32), ("push_interval", C.c_int32), ("gain_lo", C.c_float), ("gain_hi", C.c_float),
                ("friction_lo", C.c_float), ("friction_hi", C.c_float), ("push_velocity", C.c_float)]


class OptimizerT(C.Structure):
    _fields_ = [("type", C.c_int32), ("param", f32p), ("grad", f32p), ("state1", f32p), ("state2", f32p), ("count", C.c_int64), ("lr", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("step", C.c_int32), ("param16", f32p),
                ("zero_grad", C.c_int32), ("param16_planes", C.c_int32)]


class SectionT(C.Structure):
    _fields_ = [("stream", C.c_int32), ("first", C.c_int32), ("last", C.c_int32), ("wait_before", C.c_int32), ("wait_after", C.c_int32), ("bucket", C.c_int32)]


MLP_MAX_HIDDEN = 4
_H = MLP_MAX_HIDDEN


class MlpT(C.Structure):
    _fields_ = [("num_hidden", C.c_int32), ("in_dim", C.c_int32), ("in_ld", C.c_int32), ("hidden", C.c_int32 * _H), ("head_rows", C.c_int32), ("precision", C.c_int32),
                ("storage", C.c_int32), ("W", f32p * _H), ("b", f32p * _H), ("Wh", f32p), ("bh", f32p), ("gW", f32p * _H), ("gb", f32p * _H), ("gWh", f32p), ("gbh", f32p),
                ("W16", f32p * _H), ("W16t", f32p * _H), ("rows_cap", C.c_int32), ("h", f32p * _H), ("dz", f32p * _H), ("hbits", f32p * _H), ("h16", f32p * _H),
                ("dz16", f32p * _H), ("slabs", f32p), ("slab_floats", C.c_int64), ("slabs_top", f32p), ("bias_replicas", f32p), ("bias_replica_rows", C.c_int32),
                ("flat_params", f32p), ("flat_trans16", f32p),
                ("flat_count", C.c_int64), ("t_offset", C.POINTER(C.c_int64)), ("t_rows", C.POINTER(C.c_int32)), ("t_cols", C.POINTER(C.c_int32)), ("t_count", C.c_int32),
                ("amax", f32p), ("w_amax", f32p), ("deterministic", C.c_int32), ("ordered_scratch", f32p)]


class ExtraDwT(C.Structure):
    _fields_ = [("A", f32p), ("lda", C.c_int32), ("B", f32p), ("ldb", C.c_int32), ("rows", C.c_int64), ("a_amax", f32p), ("b_amax", f32p)]


class MlpMarksT(C.Structure):
    _fields_ = [("launches", C.c_int32), ("early", C.c_int32), ("dw_first", C.c_int32 * _H), ("dw_last", C.c_int32 * _H)]


class PpoLossT(C.Structure):
    _fields_ = [("actor", C.POINTER(MlpT)), ("critic", C.POINTER(MlpT)), ("rows", C.c_int32), ("norm_obs", f32p), ("norm_obs16", f32p), ("norm_obs_amax", f32p),
                ("norm_action", f32p),
                ("old_logp", f32p), ("adv", f32p), ("tar_val", f32p), ("rand_mask", f32p), ("action_std", C.c_float), ("logp_const", C.c_float),
                ("ppo_clip_ratio", C.c_float), ("action_bound_weight", C.c_float), ("action_reg_weight", C.c_float), ("critic_loss_weight", C.c_float),
                ("grad_scale", C.c_float), ("dist", f32p), ("g_logstd", f32p), ("action_entropy_weight", C.c_float), ("head_precision", C.c_int32), ("mean", f32p), ("d_mean", f32p), ("dv", f32p), ("num_valid", f32p), ("stats", f32p)]


class PpoMarksT(C.Structure):
    _fields_ = [("launches", C.c_int32), ("actor_end", C.c_int32), ("actor_early", C.c_int32), ("critic_early", C.c_int32)]


class DiscLossT(C.Structure):
    _fields_ = [("disc", C.POINTER(MlpT)), ("rows", C.c_int32), ("disc_dim", C.c_int32), ("norm_diff", f32p), ("norm_diff16", f32p), ("norm_diff_amax", f32p), ("loss_scale", C.c_float),
                ("logit_reg", C.c_float), ("grad_penalty", C.c_float), ("weight_decay", C.c_float), ("dlogit", f32p), ("a", f32p * _H), ("e", f32p * _H),
                ("g", f32p), ("G", f32p), ("a16", f32p * _H), ("e16", f32p * _H), ("G16", f32p), ("stats", f32p)]


class ActorHeadT(C.Structure):
    _fields_ = [("rows", C.c_int32), ("hidden", C.c_int32), ("H", f32p), ("Wh", f32p), ("bh", f32p), ("norm_action", f32p), ("old_logp", f32p), ("adv", f32p),
                ("rand_mask", f32p), ("n_valid", f32p), ("action_std", C.c_float), ("logp_const", C.c_float), ("clip_ratio", C.c_float),
                ("bound_weight", C.c_float), ("reg_weight", C.c_float), ("loss_scale", C.c_float), ("dist", f32p), ("dz", f32p), ("dz16", f32p), ("planes16", C.c_int32),
                ("slabs", f32p), ("num_slabs", C.c_int32), ("gb_top", f32p), ("gb_replicas", C.c_int32), ("ld_gb", C.c_int32), ("stats", f32p), ("amax", f32p)]


class DiscMarksT(C.Structure):
    _fields_ = [("launches", C.c_int32), ("head", C.c_int32), ("chain", C.c_int32), ("backward", C.c_int32), ("top_dw_first", C.c_int32), ("top_dw_last", C.c_int32)]


BWD_GRADS_ZEROED, BWD_TOP_BIAS_DONE, BWD_ACCUMULATE_DW, BWD_TOP_CAST_DONE, BWD_SIGN_BITS, BWD_TOP_BIAS_REPLICAS = 1, 2, 4, 8, 16, 32
BUCKET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.c_void_p)  # addhip_bucket_fn(user, bucket, stream)
MAX_STREAMS = 8
OPT_ADAMW, OPT_SGD = 0, 1
RIGID_BODY_W, RIGID_TOPO_W = 32, 8
EPI_NONE, EPI_BIAS, EPI_BIAS_RELU, EPI_MASK = 0, 1, 2, 3
PREC_F32, PREC_BF16, PREC_BF16X2, PREC_BF16X3, PREC_F16X2 = 0, 1, 2, 3, 4
ORDERED_BLOCKS, HEAD_BWD_BLOCKS = 64, 256  # fixed-order reductions (agent.deterministic)
DIST_FLOATS = 64                           # addhip_dist_refresh's vector (actor_std_type CONSTANT)


def actor_head_slab(hidden):
    """ADDHIP_ACTOR_HEAD_SLAB: floats of one workgroup's slab of addhip_actor_head."""
    return 32 * hidden + 64
AMAX_SLOTS, MLP_AMAX_TENSORS = 64, 16  # tracked operand maxima of PREC_F16X2 (include/addhip.h)
STORE_BF16, STORE_BF16X3 = 1, 3  # 16-bit storage formats of GEMM operands (include/addhip.h, "plane storage")
GEMM_HINT_BIG_TILE, GEMM_HINT_NO_BIG_TILE, GEMM_HINT_ONE_STAGE, GEMM_HINT_TWO_STAGE, GEMM_HINT_REG_STAGED, GEMM_HINT_WIDE_TILE = 1, 2, 4, 8, 16, 32
GEMM_MAX_GROUP = 4

i32, i64, u64, f32, vp = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_void_p
P = C.POINTER

# name -> argtypes (all return int).  Must list every symbol include/addhip.h declares.
SIGNATURES = {
    "addhip_env_step": [P(MotionT), P(TaskT), P(EnvT), P(StepOutT), i32, vp],
    "addhip_env_reset": [P(MotionT), P(TaskT), P(EnvT), P(SamplerT), vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "addhip_motion_lookup": [P(MotionT), vp, vp, i32, vp, vp, vp, vp],
    "addhip_kin_engine_step": [vp, vp, vp, i32, i32, f32, f32, vp],
    "addhip_rigid_step": [P(RigidModelT), vp, vp, vp, i32, i32, vp, vp, vp],
    "addhip_rigid_randomize": [P(RigidDrT), vp, vp, i32, vp, i32, vp],
    "addhip_gemm_f32": [P(GemmT), vp],
    "addhip_gemm_grouped": [P(GemmT), i32, vp],
    "addhip_to_bf16": [vp, vp, i64, i32, i32, i32, vp],
    "addhip_normalize_to_bf16": [vp, vp, vp, vp, i64, i32, i32, i32, vp],
    "addhip_to_bf16x3": [vp, vp, i64, i32, i32, i32, vp],
    "addhip_to_bf16_t": [vp, vp, i32, i32, i32, i32, vp],
    "addhip_shadow_refresh": [vp, vp, vp, i64, vp, vp, vp, i32, i32, vp],
    "addhip_amax_f32": [vp, i64, vp, vp],
    "addhip_slab_reduce": [vp, i32, i64, vp, i64, f32, i32, vp],
    "addhip_slab_reduce_pair": [vp, i32, i64, vp, i64, f32, i32, vp, i32, i32, vp, i32, i32, i32, vp],
    "addhip_col_sum": [vp, i32, i32, i32, vp, f32, i32, vp],
    "addhip_col_sum_ordered": [vp, i32, i32, i32, vp, f32, i32, vp, vp],
    "addhip_dist_refresh": [vp, vp, vp],
    "addhip_actor_sample": [vp, i32, vp, f32, f32, vp, vp, vp, vp, i32, i32, vp, f32, vp, vp, vp, vp],
    "addhip_fill_normal": [vp, i64, u64, u64, vp],
    "addhip_fill_uniform": [vp, i64, u64, u64, vp],
    "addhip_fill_normal_at": [vp, i64, u64, u64, vp, vp],
    "addhip_fill_uniform_at": [vp, i64, u64, u64, vp, vp],
    "addhip_fill_zero": [vp, i64, vp],
    "addhip_disc_prep": [vp, vp, i32, i32, i64, vp, f32, vp, vp, vp, P(SamplerT), i32, vp, vp],
    "addhip_sampler_update": [P(SamplerT), i32, vp],
    "addhip_disc_reward": [vp, vp, i64, f32, f32, f32, vp, vp],
    "addhip_head_gemv": [vp, i32, i32, i64, vp, vp, vp, vp],
    "addhip_td_lambda_adv": [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, f32, f32, vp, vp, vp, vp, vp],
    "addhip_norm_accum": [vp, i64, i32, i32, vp, vp, vp],
    "addhip_norm_merge": [vp, vp, vp, vp, vp, vp, i64, i32, f32, i32, vp],
    "addhip_diffnorm_merge": [vp, vp, vp, i64, i32, vp],
    "addhip_gather_minibatch": [P(GatherT), vp],
    "addhip_actor_loss": [vp, vp, vp, vp, vp, i32, f32, f32, vp, f32, f32, f32, f32, vp, vp, vp, vp, i32, vp, f32, vp],
    "addhip_count_mask": [vp, i32, vp, vp],
    "addhip_actor_head_slabs": [i32],     # returns the slab count
    "addhip_actor_head": [P(ActorHeadT), vp],
    "addhip_critic_head": [vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp],
    "addhip_disc_head": [vp, i32, i32, i32, vp, vp, vp, f32, vp, vp, vp, vp],
    "addhip_head_backward": [vp, vp, vp, i32, i32, i64, vp, vp, i32, vp, vp, vp, vp, vp, vp],
    "addhip_outer_mask": [vp, vp, vp, i32, i32, i64, vp, vp],
    "addhip_bcast_mask": [vp, vp, i32, i32, i64, vp, vp, i32, vp, vp],
    "addhip_grad_penalty": [vp, i32, i32, i32, f32, vp, vp, i32, vp, vp, vp],
    "addhip_weighted_col_sum": [vp, vp, i32, i32, i64, vp, f32, i32, vp],
    "addhip_l2_grad": [vp, vp, i64, f32, vp, vp],
    "addhip_grad_clip": [vp, i64, f32, vp, vp, vp],
    "addhip_adamw": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp],
    "addhip_sgd": [vp, vp, vp, i64, f32, f32, f32, i32, vp],
    "addhip_optimizer_step": [P(OptimizerT), vp],
    "addhip_return_tracker_fold": [vp, i32, vp, vp],
    # recorded plans / schedules (host-side runtime; handles are opaque pointers)
    "addhip_plan_create": [P(vp)],
    "addhip_plan_destroy": [vp],
    "addhip_plan_record_begin": [vp],
    "addhip_plan_record_end": [vp],
    "addhip_plan_size": [vp],                      # returns the number of calls
    "addhip_plan_call_name": [vp, i32],            # returns const char*
    "addhip_plan_call_gemms": [vp, i32, P(GemmT), i32],  # returns the number of descriptors
    "addhip_plan_run": [vp, i32, i32, vp],
    "addhip_schedule_create": [vp, P(SectionT), i32, i32, P(vp)],
    "addhip_schedule_destroy": [vp],
    "addhip_schedule_run": [vp, P(vp), BUCKET_FN, vp],
    # composite entry points (the library assembles the launches)
    "addhip_mlp_forward": [P(MlpT), vp, vp, i64, vp, vp, i32, vp, vp],
    "addhip_mlp_backward": [P(MlpT), vp, vp, i64, P(ExtraDwT), i32, P(MlpMarksT), vp, vp],
    "addhip_ppo_loss_fwd_bwd": [P(PpoLossT), P(PpoMarksT), vp],
    "addhip_disc_loss_fwd_bwd": [P(DiscLossT), P(DiscMarksT), vp],
    "addhip_update_schedule": [i32, P(PpoMarksT), P(DiscMarksT), P(SectionT), i32],   # returns the number of sections
}

_lib = None


def load():
    """Load libaddhip.so (in-tree, built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: libaddhip.so must bind to the HIP runtime already in the process (the one torch ships and allocates
    # with); loaded the other way round, two runtimes coexist and launches fail with "no ROCm-capable device"
    import torch  # noqa: F401

    if not os.path.exists(LIB_PATH):
        raise AddhipError(f"{LIB_PATH} not found: build it with `make -C add-gym_amd/csrc` (or __graft_entry__.build()). "
                          "There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    lib.addhip_last_error.restype = C.c_char_p
    lib.addhip_version.restype = C.c_int
    lib.addhip_abi_sizes.argtypes, lib.addhip_abi_sizes.restype = [C.POINTER(C.c_int32), C.c_int32], C.c_int
    structs = (MotionT, TaskT, EnvT, StepOutT, SamplerT, GemmT, GatherT, RigidModelT, RigidDrT, OptimizerT, SectionT, MlpT, ExtraDwT, MlpMarksT, PpoLossT,
               PpoMarksT, DiscLossT, DiscMarksT, ActorHeadT)
    sizes = (C.c_int32 * len(structs))()
    mine = [C.sizeof(t) for t in structs]
    if lib.addhip_abi_sizes(sizes, len(structs)) != len(structs) or list(sizes) != mine:
        raise AddhipError(f"struct layouts of this binding {mine} do not match {LIB_PATH} {list(sizes)}: rebuild the library")
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = args
        fn.restype = C.c_int
    lib.addhip_plan_call_name.restype = C.c_char_p
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise AddhipError(f"{name} failed ({rc}): {lib.addhip_last_error().decode()}")


def ptr(t):
    """Device (or host) address of a torch tensor; None -> NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "addhip buffers must be contiguous"
    return t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream

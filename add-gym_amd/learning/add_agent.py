"""ADDAgent: rollout + PPO/ADD update loop on one MI355X, same surface as the reference's
add_gym.learning.add.add_agent.ADDAgent (constructed from the Hydra tree, .train_model / .test_model
/ .save / .load; main.py:73-112), with every per-step and per-minibatch computation running in
libaddhip (HIP) on device-resident buffers.  No host synchronisation inside an iteration except one
read of the logged scalars at its end.

Restates (host orchestration only): BaseAgent.train_model/_train_iter/_rollout_train
(base_agent.py:79-114, 353-391), PPOAgent._build_train_data/_update_model (ppo_agent.py:111-192),
ADDAgent._build_train_data/_compute_disc_loss/_step_env/_reset_envs (add_agent.py:110-233).
"""
import gc
import math
import time

import numpy as np
import torch

from .. import _lib as L
from .. import dist as D
from ..anim.motion_lib import MotionLib
from ..envs.env import ImitationEnvironment
from .. import hotpath as H
from ..hotpath import gemm, make_task
from .agent_io import AgentIO
from .model import Model, NetRunner, Plan, Schedule


_SIDE_STREAMS = {}


def _side_streams(dev):
    """The three side streams of the update step, created ONCE per process and device and shared by every agent: HIP maps streams
    onto a few hardware queues in creation order, so a second agent's fresh streams can land on the queue of the main stream
    and silently serialise against it (measured: the same bf16 iteration 54 ms as the first agent of a process, 79 ms as the
    second).  The discriminator's section is the step's longest chain: its two streams get the higher priority."""
    key = torch.device(dev).index or 0
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1), torch.cuda.Stream(device=dev, priority=-1)]
    return _SIDE_STREAMS[key]


class AgentMode:
    TRAIN, TEST = 0, 1


class ADDAgent(AgentIO):
    NAME = "ADD"

    def __init__(self, env_config, distributed=False):
        if not torch.cuda.is_available():
            raise RuntimeError("ADDAgent needs a GPU: the hot path is HIP-only (no CPU fallback)")
        L.load()
        # one process per GPU: the rank's device is torch's current device (set by the launcher before construction;
        # with the reference's device-masking launch pattern, main.py:143-158, that is logical device 0)
        self._device = torch.device("cuda", torch.cuda.current_device())
        dev = self._device
        self._cfg = env_config
        cfg = self._config = env_config["agent"]
        task = self._task_cfg = env_config["task"]
        self._distributed = bool(distributed) and torch.distributed.is_available() and torch.distributed.is_initialized()
        self._world = torch.distributed.get_world_size() if self._distributed else 1
        self._rank = torch.distributed.get_rank() if self._distributed else 0
        self._seed = int(env_config.get("seed", 0)) * 1000003 + self._rank

        self._env = ImitationEnvironment(env_config, dev)
        env = self._env
        self.N = N = env.num_envs
        self._load_params(cfg)
        self.T = T = self._steps_per_iter
        # agent.matmul_precision: "fp32" (fp32 MFMA, default), "bf16x3" (exact 3-way bf16 split, fp32-level error, ~2.7x the
        # MFMA rate), "bf16x2" (two leading chunks, TF32-class) or "bf16"; operands, results and every other kernel stay fp32
        # (include/addhip.h: ADDHIP_PREC_*)
        prec = str(cfg.get("matmul_precision", "fp32"))
        if prec not in H.PRECISIONS:
            raise ValueError(f"agent.matmul_precision must be one of {sorted(H.PRECISIONS)}")
        self._matmul_precision = prec       # a property of this agent: carried by every GEMM descriptor it builds
        self._prec = H.PRECISIONS[prec]
        # "bf16" = bf16 STORAGE in the update step: hidden activations, pre-activation gradients and a shadow of the weights live
        # in HBM as bf16 (fp32 master weights, fp32 accumulation, AdamW in fp32); the head / loss kernels, the rollout and the
        # value / discriminator evaluation passes keep fp32 operands (formed by the bf16x2 products)
        # "bf16x3_planes" = PLANE STORAGE in the update step: the same buffers hold three bf16 per value whose sum is the fp32 value exactly,
        # written once by their producers; the GEMMs form the six bf16 products of the exact split from them (csrc/gemm_x3.hip): the error of
        # "bf16x3", whose GEMMs split their fp32 operands on the fly (gemm_split.hip) -- as this mode's rollout / evaluation passes do.
        self._storage16 = {"bf16": L.STORE_BF16, "bf16x3_planes": L.STORE_BF16X3}.get(prec, 0)
        self._prec_small = H.PRECISIONS["bf16x2"] if self._storage16 == L.STORE_BF16 else self._prec
        # agent.rollout_precision: the products of the rollout / value / discriminator-reward passes where they differ from the update
        # step's -- i.e. in bf16-storage mode, whose rollout keeps fp32 operands.  Default bf16x2 (16-bit operands: actions within 5e-5 of
        # the fp32 rollout); "bf16" = one bf16 product per term on operands cut to 8 bits on the way into LDS (what the update step's
        # storage holds anyway); "bf16_storage" = those passes on bf16 STORAGE like the update step -- a normalise-and-round pass per batch of
        # observations, then the storage GEMMs on the weight shadow (actions then carry the 8-bit operand rounding, ~1e-3: configs[2]-style
        # throughput runs; tests/test_hip_fullsize.py states the tolerance)
        rp = cfg.get("rollout_precision", None)
        if rp is not None and str(rp) not in ("bf16x2", "bf16", "bf16x3", "fp32", "bf16_storage"):
            raise ValueError("agent.rollout_precision must be one of bf16x2, bf16, bf16x3, fp32, bf16_storage (or null: the mode's own)")
        self._roll_storage = self._storage16 == L.STORE_BF16 and str(rp) == "bf16_storage"
        self._prec_roll = self._prec_small if rp is None or self._storage16 != L.STORE_BF16 or self._roll_storage else H.PRECISIONS[str(rp)]

        # ---- motion library + sampler (add_motion.py:14-33)
        kin = env.robot._kin_char_model
        self._motion_lib = MotionLib(task["motion_file"], list(task["motion_joint_order"]), kin, env.ctrl_dt, dev,
                                     reference_compat=task.get("reference_compat", True), cache_dir=task.get("motion_cache_dir", None))
        lib = self._motion_lib
        self._task = make_task(task, env.ctrl_dt, max_episode_length=lib.get_total_length())
        tk = self._task
        if tk.max_episode_length < T * env.ctrl_dt:
            # obs_timeout keeps ONE pre-reset row per env and iteration (include/addhip.h: addhip_td_lambda_adv)
            raise NotImplementedError("task.max_episode_length shorter than one rollout (steps_per_iter * dt) is not supported")
        scfg = task.get("sampler", {}) or {}
        self._num_segments = int(scfg.get("num_segments", 20))
        C = lib.get_num_motions()
        self._num_clips = C
        seg = (lib.get_motion_lengths() / self._num_segments).to(dev)  # sampler.py:13-15
        cdf = torch.cumsum(lib.get_motion_weights(), 0).to(dev)
        self._smp = dict(errors=torch.ones(C, self._num_segments, device=dev), seg=seg, cdf=cdf,
                         bits=torch.zeros(1, dtype=torch.int32, device=dev), esum=torch.zeros(C * self._num_segments, device=dev),
                         ecnt=torch.zeros(C * self._num_segments, device=dev))
        temp = scfg.get("temperature", None)
        self._smp_c = L.SamplerT(L.ptr(self._smp["errors"]), L.ptr(seg), L.ptr(cdf), self._num_segments, -1.0 if temp is None else float(temp),
                                 float((task.get("num_disc_obs_steps", 1) - 1) * env.ctrl_dt), int(bool(task.get("rand_reset", True))),
                                 L.ptr(self._smp["bits"]), L.ptr(self._smp["esum"]), L.ptr(self._smp["ecnt"]))

        # ---- per-env state
        OS, DS = tk.obs_stride, tk.disc_stride
        self._obs_dim, self._disc_dim = tk.obs_dim, tk.disc_dim
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        ent = env.robot.entity
        self._fast_engine = hasattr(ent, "hot_state")
        if self._fast_engine:
            if hasattr(ent, "set_termination_links"):  # the engine evaluates the non-foot ground-contact predicate itself
                ent.set_termination_links(list(task.get("contact_bodies", [])))
            sim_pose, sim_vel, contact = ent.hot_state()
        else:
            sim_pose, sim_vel, contact = z(N, L.POSE_W), z(N, L.POSE_W), z(N, dt=torch.uint8)
        if not self._fast_engine:  # contact predicate inputs (add_done.py:36-57)
            names = list(task.get("contact_bodies", []))
            allowed = [ent.get_link(name=nm).idx for nm in names]
            self._noncontact_ids = torch.tensor([l.idx for l in ent.links if l.idx not in allowed], dtype=torch.long, device=dev)
        self._S = S = dict(sim_pose=sim_pose, sim_vel=sim_vel, contact=contact, time=env.time_buf, time_off=z(N), motion_id=z(N, dt=torch.int32),
                           hist=z(N, tk.num_disc_obs_steps, L.POSE_W), hist_vel=z(N, tk.num_disc_obs_steps, L.POSE_W) if tk.enable_vel_obs else None, done=z(N, dt=torch.int32),
                           ref_pose=None, ref_vel=None,  # optional outputs of the step kernel (gathered reference rows); not needed here
                           ret_acc=z(N), len_acc=z(N, dt=torch.int32), ret_acc_test=z(N), len_acc_test=z(N, dt=torch.int32))
        jw = task.get("joint_err_w", None)  # add_reward.py:24-52: per joint in kinematic-tree order; every G1 joint has one dof
        S["dof_err_w"] = None if jw is None else torch.tensor([float(w) for w in jw], dtype=torch.float32, device=dev)
        self._env_c = L.EnvT(N, *[L.ptr(S[k]) for k in ("sim_pose", "sim_vel", "time", "time_off", "motion_id", "hist", "hist_vel", "done", "contact",
                                                         "ref_pose", "ref_vel", "ret_acc", "len_acc", "dof_err_w")])
        self._env_c_test = L.EnvT(N, *[L.ptr(S[k]) for k in ("sim_pose", "sim_vel", "time", "time_off", "motion_id", "hist", "hist_vel", "done", "contact",
                                                              "ref_pose", "ref_vel", "ret_acc_test", "len_acc_test", "dof_err_w")])
        self._head = 0  # ring slot that receives the next state (circular_buffer.py:8)

        # ---- experience buffer (experience_buffer.py; 13 buffers of base/ppo/amp/add agents), obs has T+1 slots
        self._B = B = dict(obs=z(T + 1, N, OS), obs_timeout=z(N, OS), action=z(T + 1, N, 32), a_logp=z(T + 1, N), rand_mask=z(T + 1, N), reward=z(T, N),
                           done=z(T, N, dt=torch.int32), disc_obs=z(T + 1, N, DS), disc_demo=z(T + 1, N, DS), motion_id=z(T, N, dt=torch.int32),
                           motion_time=z(T, N), tar_val=z(T, N), adv=z(T, N), vals=z(T + 1, N), timeout_vals=z(N), ep_stats=z(T, 3))
        self._total_samples = 0
        self._sample_count = 0
        self._iter = 0

        # ---- normalisers (base_agent.py:226-252, add_agent.py:84-91)
        lo, hi = kin.action_bounds()
        a_mean, a_std = 0.5 * (hi + lo), 0.5 * (hi - lo)
        self._a_low, self._a_high = lo, hi
        self._Nrm = dict(obs_mean=z(OS), obs_std=torch.ones(OS, device=dev), obs_msq=z(OS), obs_cnt=z(1, dt=torch.int64), obs_sum=z(OS), obs_sumsq=z(OS),
                         a_mean=torch.cat([a_mean, torch.zeros(3)]).to(dev), a_std=torch.cat([a_std, torch.ones(3)]).to(dev),
                         a_cnt=z(1, dt=torch.int64), d_abs=torch.ones(DS, device=dev), d_cnt=z(1, dt=torch.int64), d_sum=z(DS))
        self._obs_norm_first = True

        # ---- model + optimiser state (mp_optimizer.py:25-40)
        self._model = Model(cfg["model"], tk.obs_dim, OS, tk.disc_dim, DS, dev, seed=self._seed)
        opt = cfg["optimizer"]
        self._opt_type = str(opt["type"])  # mp_optimizer.py:32-40: "Adam" -> torch.optim.AdamW, "SGD" -> torch.optim.SGD(momentum=0.9)
        if self._opt_type not in ("Adam", "SGD"):
            raise ValueError("Unsupported optimizer type: " + self._opt_type)
        self._lr = float(opt["learning_rate"])
        self._wd = float(opt.get("weight_decay", 0.0))
        self._grad_clip = float(opt.get("grad_clip", 0.0))  # mp_optimizer.py:10, 42-46 (0 = off: the reference's default)
        if self._distributed:  # DDP ctor behaviour: every rank starts from rank 0's weights (base_agent.py:50-57)
            D.broadcast_(self._model.params, 0)
        if self._storage16:
            self._model.enable_shadow(self._storage16)
        # "f16x2": fp32 operands split on the fly into two fp16 planes on per-tensor power-of-two scales (csrc/gemm_split.hip, ADDHIP_PREC_F16X2);
        # the scales come from maxima the producers track on the device (activations / gradients: NetRunner.amax; parameters: Model.w_amax)
        # agent.deterministic: every reduction behind a gradient in a fixed order (no float atomics): two runs from the same state give
        # bit-identical gradients and parameters, as the reference's CPU path does under a seed (mp_optimizer.py:14-23)
        self._deterministic = bool(cfg.get("deterministic", False))
        self._f16x2 = self._prec == L.PREC_F16X2
        if self._f16x2:
            self._model.enable_w_amax()
        mm = self._model
        sgd = self._opt_type == "SGD"
        self._opt_c = L.OptimizerT(L.OPT_SGD if sgd else L.OPT_ADAMW, L.ptr(mm.params), L.ptr(mm.grads), L.ptr(mm.exp_avg), None if sgd else L.ptr(mm.exp_avg_sq),
                                   mm.count, self._lr, 0.9, 0.999, 1e-8, self._wd, 1, L.ptr(mm.params16) if self._storage16 else None, 1, self._storage16)

        self._build_workspace()
        self._build_plans()
        self._mode = AgentMode.TRAIN
        self._is_restored = False
        self._logger = None
        self._test_state = z(3)
        self._train_state = z(3)
        self._timers = {}
        self._test_calls = 0
        # agent.rollout_graph: capture the T steps of a training rollout (actor MLP, noise, engine step, fused env step, masked
        # reset: ~13 launches per step) into one hipGraph per ring phase and replay it every iteration.  The Philox stream ids
        # come from a device counter advanced once per iteration, so every replay draws fresh numbers -- the same numbers the
        # call-by-call path draws (the two are bit-identical; tests/test_hip_agent.py).
        self._rollout_graph = bool(cfg.get("rollout_graph", False))
        self._graphs, self._graph_warm, self._graph_model_version = {}, False, None
        self._sid_base = torch.zeros(1, dtype=torch.int64, device=dev)
        # optional externally supplied random draws (parity tests replay the reference's draws through these):
        #   {"noise": [T][N,29], "uniforms": {Philox stream id (stream_* below): [3,N]}, "perms": iterator of int64 permutations, "pre_step": fn(t)}
        self.inject = None

    # ------------------------------------------------------------------ config (base/ppo/amp agents' _load_params)
    def _load_params(self, c):
        self._discount = float(c["discount"])
        self._iters_per_output = int(c["iters_per_output"])
        self._normalizer_samples = c.get("normalizer_samples", math.inf)
        self._test_episodes = int(c["test_episodes"])
        self._steps_per_iter = int(c["steps_per_iter"])
        self._update_epochs = int(c["update_epochs"])
        self._batch_size = int(c["batch_size"])
        self._td_lambda = float(c["td_lambda"])
        self._ppo_clip_ratio = float(c["ppo_clip_ratio"])
        self._norm_adv_clip = float(c["norm_adv_clip"])
        self._action_bound_weight = float(c["action_bound_weight"])
        self._action_entropy_weight = float(c["action_entropy_weight"])  # ppo_agent.py:262-272
        self._action_reg_weight = float(c["action_reg_weight"])
        self._critic_loss_weight = float(c["critic_loss_weight"])
        self._exp_anneal_samples = float(c.get("exp_anneal_samples", float("inf")))  # ppo_agent.py:32-34
        self._exp_prob_beg = float(c.get("exp_prob_beg", 1.0))
        self._exp_prob_end = float(c.get("exp_prob_end", 1.0))
        self._disc_loss_weight = float(c["disc_loss_weight"])
        self._disc_logit_reg = float(c["disc_logit_reg"])
        self._disc_grad_penalty = float(c["disc_grad_penalty"])
        self._disc_weight_decay = float(c["disc_weight_decay"])
        self._disc_reward_scale = float(c["disc_reward_scale"])
        self._task_reward_weight = float(c["task_reward_weight"])
        self._disc_reward_weight = float(c["disc_reward_weight"])
        self._max_samples = c.get("max_samples", int(1e6))

    # ------------------------------------------------------------------ workspace + plans
    def _build_workspace(self):
        dev, N, T, m = self._device, self.N, self.T, self._model
        self.Mb = Mb = self._batch_size * N  # ppo_agent.py:176
        self._eval_rows = min(T * N, 65536)
        rows = max(N, Mb + 1, self._eval_rows)
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        from .model import split_k_for

        # actor-head weight-gradient slabs: 32 K slices of the 32-wide GEMM, or one per workgroup of addhip_actor_head (<= 256)
        need = max(32 * m.actor.head_rows * m.actor.hidden[-1], 256 * L.actor_head_slab(m.actor.hidden[-1]))
        for net in m.nets:
            for i, h in enumerate(net.hidden):
                in_ld = net.in_ld if i == 0 else net.hidden[i - 1]
                need = max(need, 2 * split_k_for(h, in_ld, Mb + 1) * h * in_ld)  # x2: gradient-penalty product shares the reduce
        # split-K scratch: one per net, because the three nets' update sections run on three concurrent streams
        self._slabs_all = z(4, need)  # [3]: the discriminator's top-layer weight gradient, which runs beside the rest of its backward pass
        self._slabs = self._slabs_all[0]
        s16 = self._storage16
        det = self._deterministic
        self._run_actor = NetRunner(m, m.actor, Mb + 1 if s16 else rows, dev, self._slabs_all[0], self._prec, s16, det)
        self._run_critic = NetRunner(m, m.critic, Mb + 1 if s16 else rows, dev, self._slabs_all[1], self._prec, s16, det)
        self._run_disc = NetRunner(m, m.disc, Mb + 1 if s16 else rows, dev, self._slabs_all[2], self._prec, s16, det)
        self._run_disc.aux_slabs = (len(m.disc.hidden) - 1, self._slabs_all[3])
        # rollout / evaluation passes: the same runners, except in bf16-storage mode (fp32 operands, bf16x2 products)
        if self._roll_storage:
            self._roll_actor = NetRunner(m, m.actor, N, dev, None, L.PREC_BF16, L.STORE_BF16)
            self._eval_critic = NetRunner(m, m.critic, self._eval_rows, dev, None, L.PREC_BF16, L.STORE_BF16)
            self._eval_disc = NetRunner(m, m.disc, self._eval_rows, dev, None, L.PREC_BF16, L.STORE_BF16)
        elif s16:
            self._roll_actor = NetRunner(m, m.actor, N, dev, None, self._prec_roll)
            self._eval_critic = NetRunner(m, m.critic, self._eval_rows, dev, None, self._prec_roll)
            self._eval_disc = NetRunner(m, m.disc, self._eval_rows, dev, None, self._prec_roll)
        else:
            self._roll_actor, self._eval_critic, self._eval_disc = self._run_actor, self._run_critic, self._run_disc
        self._side_streams = _side_streams(dev)
        OS, DS = self._task.obs_stride, self._task.disc_stride
        hd = m.disc.hidden
        HR = m.actor.head_rows  # 32, or 64 with a log-std head (actor_std_type VARIABLE: mean | log-std columns)
        self._W = dict(mean=z(rows, HR), d_mean=z(rows, HR), noise=z(N, L.NUM_DOF), explore_u=z(N), u=z(3, N), logits=z(rows), nv=z(1),
                       norm_obs=z(Mb, OS), norm_act=z(Mb, 32), mb_logp=z(Mb), mb_adv=z(Mb), mb_tar=z(Mb), mb_mask=z(Mb), norm_diff=z(rows + 1, DS),
                       dv=z(Mb), dlogit=z(Mb + 1), g=z(Mb, DS), G=z(Mb, DS),
                       stats=z(32), scratch=z(4096, dt=torch.float64), adv_stats=z(2), rstats=z(2), perm_idx=z(Mb, dt=torch.int64),
                       mb_amax=z(2, L.AMAX_SLOTS, dt=torch.int32))  # f16x2: tracked maxima of the gathered norm_obs / norm_diff rows
        for i, h in enumerate(hd):  # the gradient-penalty chain's rows of discriminator layer i (fp32; storage modes keep only e of the last layer)
            if not s16:
                self._W[f"gp_a{i}"] = z(Mb, h)
            if not s16 or i == len(hd) - 1:
                self._W[f"gp_e{i}"] = z(Mb, h)
        if s16:  # bf16 copies of the GEMM operands the fp32 kernels produce
            b16 = lambda r, c: torch.zeros(r, s16 * c, dtype=torch.bfloat16, device=dev)  # (plane storage: three bf16 per value)
            self._W.update(norm_obs16=b16(Mb, OS), norm_diff16=b16(Mb + 1, DS), G16=b16(Mb, DS))
            for i, h in enumerate(hd):  # the penalty chain's rows of layer i as 16-bit storage (e of the last layer stays fp32)
                self._W[f"gp_a{i}_16"] = b16(Mb, h)
                if i < len(hd) - 1:
                    self._W[f"gp_e{i}_16"] = b16(Mb, h)
        if self._roll_storage:  # normalised, rounded input rows of the rollout / evaluation passes
            self._W.update(roll_x16=torch.zeros(N, OS, dtype=torch.bfloat16, device=dev), eval_x16=torch.zeros(self._eval_rows, OS, dtype=torch.bfloat16, device=dev),
                           eval_d16=torch.zeros(self._eval_rows, DS, dtype=torch.bfloat16, device=dev))

    def _gemm(self, plan, *a, **k):
        k.setdefault("precision", self._prec_roll)
        g = gemm(*a, **k)
        plan.hold(g)
        plan.add("addhip_gemm_f32", g)

    def _build_plans(self):
        m, W, B, S, Nm, tk = self._model, self._W, self._B, self._S, self._Nrm, self._task
        N, T, Mb = self.N, self.T, self.Mb
        OS, DS = tk.obs_stride, tk.disc_stride
        ra, rc, rd = self._run_actor, self._run_critic, self._run_disc
        hA, hC, hD = m.actor.hidden[-1], m.critic.hidden[-1], m.disc.hidden[-1]

        # ---- rollout step t: decide action (ppo_agent.py:72-109) -> engine -> env step (add_agent.py:204-219) -> reset
        self._act_plans, self._step_out, self._reset_args = [], [], []
        for t in range(T + 1):
            p = Plan()
            rr = self._roll_actor
            if self._roll_storage:
                p.add("addhip_normalize_to_bf16", L.ptr(B["obs"][t]), L.ptr(Nm["obs_mean"]), L.ptr(Nm["obs_std"]), L.ptr(W["roll_x16"]), N, OS, OS, OS)
                rr.forward(p, None, N, x16_ptr=L.ptr(W["roll_x16"]))
            else:
                rr.forward(p, L.ptr(B["obs"][t]), N, a_mean=L.ptr(Nm["obs_mean"]), a_std=L.ptr(Nm["obs_std"]))
            HR = m.actor.head_rows
            self._gemm(p, N, HR, hA, L.ptr(rr.h[-1]), hA, 1, m.p("actor", "Wh"), hA, 1, L.ptr(W["mean"]), HR, L.EPI_BIAS, m.p("actor", "bh"))
            self._act_plans.append(p)
        for t in range(T):
            self._step_out.append(L.StepOutT(L.ptr(B["obs"][t + 1]), None, L.ptr(B["obs_timeout"]), L.ptr(B["disc_obs"][t]), L.ptr(B["disc_demo"][t]),
                                             L.ptr(B["reward"][t]), L.ptr(B["done"][t]), L.ptr(B["motion_id"][t]), L.ptr(B["motion_time"][t]),
                                             L.ptr(B["ep_stats"][t])))

        # ---- update step on one gathered minibatch (ppo_agent.py:194-275, add_agent.py:141-202): the library records it itself.
        # addhip_ppo_loss_fwd_bwd = the actor's and the critic's sections (forward, loss heads, backward), addhip_disc_loss_fwd_bwd = the
        # discriminator's (L2 terms, forward over Mb differences + one zero-difference row, logit loss, gradient-penalty chain with its
        # second-order terms, backward); csrc/learner.hip assembles the ~55 launches, this host only hands over the buffers.
        # (the whole flat gradient is zeroed once per step, before the sections fork: _run_update_sections; bias and head
        # gradients are then accumulated by atomics from the kernels that already hold the data)
        import ctypes as C
        p = self._update_plan = Plan()
        ppo, disc = self._loss_descs()
        pm, dm = L.PpoMarksT(), L.DiscMarksT()
        p.hold(ppo, disc, self._mlp_c)
        p.add("addhip_ppo_loss_fwd_bwd", ppo, C.byref(pm))
        p.add("addhip_disc_loss_fwd_bwd", disc, C.byref(dm))
        self._update_marks = [("actor", pm.actor_end), ("critic", pm.launches), ("disc", pm.launches + dm.launches)]
        # Launch / exchange schedule of one optimiser step (addhip_update_schedule): ten sections on four streams -- actor | critic |
        # discriminator x 2 -- with event dependencies; actor and critic hand over everything but their first layers as soon as it is final
        # (buckets 0, 1), the discriminator after its two-stream backward (bucket 2); the two first layers are one bucket after the join.
        secs = (L.SectionT * 10)()
        if L.load().addhip_update_schedule(0, C.byref(pm), C.byref(dm), secs, 10) != 10:  # (returns the number of sections)
            raise L.AddhipError("addhip_update_schedule: " + L.load().addhip_last_error().decode())
        br = m.bucket_ranges
        # the plan and its schedule live in the library (include/addhip.h, "recorded plans"): one optimiser step's sections are ONE C
        # call (addhip_schedule_run: launches, stream forks / joins by HIP events, bucket call-backs), for this host or any other
        self._schedule = Schedule(p, secs, 1 + len(self._side_streams), buckets=[br["actor_tail"], br["critic_tail"], br["disc"], br["first_layers"]])

        # ---- build-train-data: critic over the T+1 obs slots and the N obs_timeout rows, discriminator over the T*N
        # differences, in chunks of _eval_rows rows (plans prebuilt once like the act / update plans)
        chunk = self._eval_rows
        self._critic_eval = []  # (plan, rows, value destination)
        for src, dst, total in ((B["obs"], B["vals"], (T + 1) * N), (B["obs_timeout"], B["timeout_vals"], N)):
            for r0 in range(0, total, chunk):
                rows = min(chunk, total - r0)
                p = Plan()
                ec = self._eval_critic
                if self._roll_storage:
                    p.add("addhip_normalize_to_bf16", L.ptr(src) + 4 * r0 * OS, L.ptr(Nm["obs_mean"]), L.ptr(Nm["obs_std"]), L.ptr(W["eval_x16"]), rows, OS, OS, OS)
                    ec.forward(p, None, rows, x16_ptr=L.ptr(W["eval_x16"]))
                else:
                    ec.forward(p, L.ptr(src) + 4 * r0 * OS, rows, a_mean=L.ptr(Nm["obs_mean"]), a_std=L.ptr(Nm["obs_std"]))
                p.add("addhip_head_gemv", L.ptr(ec.h[-1]), hC, hC, rows, m.p("critic", "Wh"), m.p("critic", "bh"), L.ptr(dst) + 4 * r0)
                self._critic_eval.append(p)
        self._disc_eval = []  # (r0, rows, forward + logits plan)
        for r0 in range(0, T * N, chunk):
            rows = min(chunk, T * N - r0)
            p = Plan()
            ed = self._eval_disc
            if self._roll_storage:
                p.add("addhip_to_bf16", L.ptr(W["norm_diff"]), L.ptr(W["eval_d16"]), rows, DS, DS, DS)
                ed.forward(p, None, rows, x16_ptr=L.ptr(W["eval_d16"]))
            else:
                ed.forward(p, L.ptr(W["norm_diff"]), rows)
            p.add("addhip_head_gemv", L.ptr(ed.h[-1]), hD, hD, rows, m.p("disc", "Wh"), m.p("disc", "bh"), L.ptr(W["logits"]))
            self._disc_eval.append((r0, rows, p))

        s16 = self._storage16
        self._gather_c = L.GatherT(L.ptr(W["perm_idx"]), Mb, L.ptr(B["obs"]), OS, tk.obs_dim, L.ptr(Nm["obs_mean"]), L.ptr(Nm["obs_std"]), L.ptr(B["action"]),
                                   L.ptr(Nm["a_mean"]), L.ptr(Nm["a_std"]), L.ptr(B["a_logp"]), L.ptr(B["adv"]), L.ptr(B["tar_val"]), L.ptr(B["rand_mask"]),
                                   L.ptr(B["disc_obs"]), L.ptr(B["disc_demo"]), DS, tk.disc_dim, L.ptr(Nm["d_abs"]), 1e-4, L.ptr(W["norm_obs"]),
                                   L.ptr(W["norm_act"]), L.ptr(W["mb_logp"]), L.ptr(W["mb_adv"]), L.ptr(W["mb_tar"]), L.ptr(W["mb_mask"]), L.ptr(W["norm_diff"]),
                                   L.ptr(W["norm_obs16"]) if s16 else None, L.ptr(W["norm_diff16"]) if s16 else None, s16,
                                   L.ptr(W["mb_amax"][0]) if self._f16x2 else None, L.ptr(W["mb_amax"][1]) if self._f16x2 else None)

    def _loss_descs(self):
        """The update step's two loss sections as parameter blocks of the composite entry points (include/addhip.h:
        addhip_ppo_loss_fwd_bwd, addhip_disc_loss_fwd_bwd) over this agent's buffers."""
        m, W, tk, Mb = self._model, self._W, self._task, self.Mb
        s16 = self._storage16
        gs = 1.0 / self._world  # every loss coefficient carries 1/world: the all-reduce SUM of the gradients is then their mean over ranks
        self._mlp_c = {r.net.name: r.c_struct() for r in (self._run_actor, self._run_critic, self._run_disc)}
        import ctypes as C
        ppo = L.PpoLossT(C.pointer(self._mlp_c["actor"]), C.pointer(self._mlp_c["critic"]), Mb, L.ptr(W["norm_obs"]), L.ptr(W["norm_obs16"]) if s16 else None,
                         L.ptr(W["mb_amax"][0]) if self._f16x2 else None, L.ptr(W["norm_act"]), L.ptr(W["mb_logp"]), L.ptr(W["mb_adv"]), L.ptr(W["mb_tar"]), L.ptr(W["mb_mask"]), m.std32, m.logp_const,
                         self._ppo_clip_ratio, self._action_bound_weight, self._action_reg_weight, self._critic_loss_weight, gs,
                         m.dist_ptr(), m.g("actor", "logstd") if m.dist is not None else None, self._action_entropy_weight, self._prec_small,
                         L.ptr(W["mean"]), L.ptr(W["d_mean"]), L.ptr(W["dv"]), L.ptr(W["nv"]), L.ptr(W["stats"]))
        o = (lambda k: L.ptr(W[k])) if s16 else (lambda k: None)
        f = (lambda k: None) if s16 else (lambda k: L.ptr(W[k]))
        disc = L.DiscLossT()
        disc.disc, disc.rows, disc.disc_dim = C.pointer(self._mlp_c["disc"]), Mb, tk.disc_dim
        disc.norm_diff, disc.norm_diff16, disc.norm_diff_amax = L.ptr(W["norm_diff"]), o("norm_diff16"), L.ptr(W["mb_amax"][1]) if self._f16x2 else None
        disc.loss_scale, disc.logit_reg, disc.grad_penalty, disc.weight_decay = self._disc_loss_weight * gs, self._disc_logit_reg, self._disc_grad_penalty, self._disc_weight_decay
        disc.dlogit, disc.g, disc.G, disc.G16, disc.stats = L.ptr(W["dlogit"]), L.ptr(W["g"]), f("G"), o("G16"), L.ptr(W["stats"])
        nd = len(m.disc.hidden)
        for i in range(nd):  # the gradient-penalty chain's workspace, per hidden layer (include/addhip.h: addhip_disc_loss_t)
            disc.a[i], disc.a16[i] = f(f"gp_a{i}"), o(f"gp_a{i}_16")
            disc.e[i] = L.ptr(W[f"gp_e{i}"]) if (not s16 or i == nd - 1) else None
            disc.e16[i] = o(f"gp_e{i}_16") if i < nd - 1 else None
        return ppo, disc

    # ------------------------------------------------------------------ public surface
    def get_num_envs(self):
        return self.N

    def calc_num_params(self):
        return self._model.num_params()

    def set_mode(self, mode):
        assert mode in (AgentMode.TRAIN, AgentMode.TEST)
        self._mode = mode

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    # ------------------------------------------------------------------ env reset / step
    # Philox stream ids (seed = agent seed): one 2^40-wide namespace per purpose, so that no two draws of a run share a
    # counter range: 1 actor noise, 2 masked train resets, 3 exploration mask, 4 reset-all (start / after an output
    # iteration), 5 reset-all at the start of an evaluation, 6 masked resets inside an evaluation.
    @staticmethod
    def stream_train_reset(step_index):
        return (2 << 40) + int(step_index)

    @staticmethod
    def stream_reset_all(count):
        return (4 << 40) + int(count)

    @staticmethod
    def stream_test_reset_all(test_call):
        return (5 << 40) + int(test_call)

    @staticmethod
    def stream_test_reset(test_call, k):
        return (6 << 40) + (int(test_call) << 20) + int(k)

    def _draw_uniforms(self, stream_id, relative=False):
        """relative: stream_id is an offset from the device counter _sid_base (hipGraph-captured rollouts)."""
        if self.inject is not None and "uniforms" in self.inject:
            self._W["u"].copy_(self.inject["uniforms"][stream_id])
            return
        if relative:
            L.call("addhip_fill_uniform_at", L.ptr(self._W["u"]), 3 * self.N, self._seed, stream_id, L.ptr(self._sid_base), self._stream())
        else:
            L.call("addhip_fill_uniform", L.ptr(self._W["u"]), 3 * self.N, self._seed, stream_id, self._stream())

    def _sync_foreign_engine_in(self):
        """Slow path for engines without hot_state(): gather the BaseEntity getters into the packed rows."""
        r, S = self._env.robot, self._S
        S["sim_pose"].copy_(torch.cat([r.base_pos, r.base_quat, r.dof_pos], dim=-1))
        S["sim_vel"][:, :35].copy_(torch.cat([r.base_lin_vel, r.base_ang_vel, r.dof_vel], dim=-1))
        if hasattr(self, "_noncontact_ids"):
            S["contact"].copy_(r.get_ground_contact_forces_v2(self._env.plane, self._noncontact_ids).to(torch.uint8))

    def _sync_foreign_engine_out(self, reset_mask):
        ids = reset_mask.nonzero(as_tuple=False).flatten()
        if len(ids) > 0:
            ent = self._env.robot.entity
            ent.set_qpos(self._S["sim_pose"][ids], envs_idx=ids)            # add_observation.py:314-322
            ent.set_dofs_velocity(self._S["sim_vel"][ids, :35], envs_idx=ids)  # :323-331

    def _reset_envs(self, reset_all, obs_slot, disc_slot, demo_slot, stream_id, relative=False):
        """ADDAgent._reset_envs (add_agent.py:221-233), masked on device (no host nonzero)."""
        S = self._S
        mask = None
        if not self._fast_engine:
            mask = torch.ones(self.N, dtype=torch.bool, device=self._device) if reset_all else (S["done"] != 0)
        self._draw_uniforms(stream_id, relative)
        u = self._W["u"]
        L.call("addhip_env_reset", self._motion_lib.c_struct, self._task, self._env_c, self._smp_c, L.ptr(u[0]), L.ptr(u[1]), L.ptr(u[2]),
               L.ptr(obs_slot), L.ptr(disc_slot), L.ptr(demo_slot), int(reset_all), self._head, self._stream())
        if not self._fast_engine:
            self._sync_foreign_engine_out(mask)

    def _decide_action(self, t, slot_t, deterministic, relative=False):
        """PPOAgent._decide_action (ppo_agent.py:72-104) + record (ppo_agent.py:106-109)."""
        B, W, Nm, m = self._B, self._W, self._Nrm, self._model
        st = self._stream()
        self._act_plans[slot_t].run(st)
        if not deterministic and self.inject is not None and "noise" in self.inject:
            W["noise"].copy_(self.inject["noise"][t])
        elif not deterministic and relative:
            L.call("addhip_fill_normal_at", L.ptr(W["noise"]), self.N * L.NUM_DOF, self._seed, (1 << 40) + t, L.ptr(self._sid_base), st)
        elif not deterministic:
            L.call("addhip_fill_normal", L.ptr(W["noise"]), self.N * L.NUM_DOF, self._seed, (1 << 40) + self._iter * self.T + t, st)
        explore_u, exp_prob = None, 1.0
        if not deterministic:
            exp_prob = self._get_exp_prob()
            if exp_prob < 1.0:  # rand_action_mask = bernoulli(exp_prob) (ppo_agent.py:80-88)
                if self.inject is not None and "explore_u" in self.inject:
                    W["explore_u"].copy_(self.inject["explore_u"][t])
                else:
                    L.call("addhip_fill_uniform", L.ptr(W["explore_u"]), self.N, self._seed, (3 << 40) + self._iter * self.T + t, st)
                explore_u = L.ptr(W["explore_u"])
        HR = m.actor.head_rows
        L.call("addhip_actor_sample", L.ptr(W["mean"]), HR, L.ptr(W["noise"]), m.std32, m.logp_const, m.dist_ptr(),
               L.ptr(W["mean"]) + 4 * 32 if HR == 64 else None, L.ptr(Nm["a_mean"]), L.ptr(Nm["a_std"]), self.N,
               int(deterministic), explore_u, exp_prob, L.ptr(B["action"][slot_t]), L.ptr(B["a_logp"][slot_t]), L.ptr(B["rand_mask"][slot_t]), st)

    def _get_exp_prob(self):
        """ppo_agent.py:161-168."""
        if math.isfinite(self._exp_anneal_samples):
            l = min(max(float(self._sample_count) / self._exp_anneal_samples, 0.0), 1.0)
            return (1.0 - l) * self._exp_prob_beg + l * self._exp_prob_end
        return self._exp_prob_beg

    def _step_env(self, slot_t, out_c, env_c):
        """Environment.step (env.py:150-155) then the fused HIP env step."""
        env = self._env
        env.robot.apply_action(self._B["action"][slot_t])
        env.scene.step()
        if not self._fast_engine:
            self._sync_foreign_engine_in()
        L.call("addhip_env_step", self._motion_lib.c_struct, self._task, env_c, out_c, self._head, self._stream())
        self._head = (self._head + 1) % self._task.num_disc_obs_steps

    # ------------------------------------------------------------------ training loop
    def _rollout_train(self):
        B, T = self._B, self.T
        if self._rollout_graph and self._graph_ok():
            self._rollout_train_graph()
            self._total_samples += T * self.N
            return
        B["ep_stats"].zero_()
        for t in range(T):
            self._decide_action(t, t, False)
            if self.inject is not None and "pre_step" in self.inject:
                self.inject["pre_step"](t)
            self._step_env(t, self._step_out[t], self._env_c)
            # obs slot t+1 already holds the post-step obs; reset envs overwrite theirs (base_agent.py:449-453)
            self._reset_envs(False, B["obs"][t + 1], B["disc_obs"][t + 1], B["disc_demo"][t + 1], self.stream_train_reset(self._iter * T + t))
        self._total_samples += T * self.N

    def _graph_ok(self):
        """A rollout can be replayed from a graph when nothing in it depends on host state that changes between iterations:
        engine state shared in place, no injected draws, constant exploration probability.  (The rigid engine's domain
        randomisation is drawn on the device from a device-resident step counter, addhip_rigid_randomize: it is captured too.)"""
        return self._fast_engine and self.inject is None and not math.isfinite(self._exp_anneal_samples) and self._exp_prob_beg >= 1.0

    def _rollout_body_relative(self):
        """The T steps with stream ids relative to the device counter (what the graph captures; also run eagerly once)."""
        B, T = self._B, self.T
        B["ep_stats"].zero_()
        for t in range(T):
            self._decide_action(t, t, False, relative=True)
            self._step_env(t, self._step_out[t], self._env_c)
            self._reset_envs(False, B["obs"][t + 1], B["disc_obs"][t + 1], B["disc_demo"][t + 1], self.stream_train_reset(t), relative=True)

    def _rollout_train_graph(self):
        T = self.T
        self._sid_base.fill_(self._iter * T)  # outside the graph: the only per-iteration input of the rollout
        h0 = self._head
        # the engine's model (gains, termination links) may have been changed since the last capture: refresh its device tables in
        # place now -- never inside a capture -- and drop graphs whose launches carry the old by-value fields
        ent = self._env.robot.entity
        if getattr(ent, "_dirty", False):
            ent._upload()
        ver = getattr(ent, "model_version", 0)
        if ver != self._graph_model_version:
            self._graphs.clear()
            self._graph_model_version = ver
        if not self._graph_warm:  # first use: run eagerly so that every kernel is loaded before a capture starts
            self._rollout_body_relative()
            self._graph_warm = True
            return
        g = self._graphs.get(h0)
        if g is None:  # one graph per ring phase (the ring slot of step t is (h0 + t) % ring depth, a launch argument)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._rollout_body_relative()
            self._graphs[h0] = g
        self._head = h0
        g.replay()
        self._head = (h0 + T) % self._task.num_disc_obs_steps

    def _build_train_data(self):
        """add_agent.py:110-139 then ppo_agent.py:111-159."""
        B, W, Nm, m, tk = self._B, self._W, self._Nrm, self._model, self._task
        T, N = self.T, self.N
        rows_total = T * N
        st = self._stream()
        W["rstats"].zero_()
        record = self._need_normalizer_update()
        # the critic pass (side stream) and the discriminator pass (this stream) are independent until TD(lambda)
        main, side = torch.cuda.current_stream(), self._side_streams[0]
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)
        cs = side.cuda_stream

        # One critic pass over the T+1 obs slots gives V(obs[t]) and, shifted by one slot, V(next_obs[t]) = V(obs[t+1]) for
        # every sample whose env was not reset (the reference evaluates next_obs separately, ppo_agent.py:117-126; the
        # rows are identical, so are the values).  Reset envs: SUCC/FAIL take the terminal values, TIME takes V of the
        # pre-reset row the step kernel parked in obs_timeout.
        for p in self._critic_eval:
            p.run(cs)
        joined = torch.cuda.Event()
        joined.record(side)
        for r0, rows, plan in self._disc_eval:
            do = L.ptr(B["disc_obs"]) + 4 * r0 * tk.disc_stride
            dd = L.ptr(B["disc_demo"]) + 4 * r0 * tk.disc_stride
            L.call("addhip_disc_prep", do, dd, tk.disc_stride, tk.disc_dim, rows, L.ptr(Nm["d_abs"]), 1e-4, L.ptr(W["norm_diff"]),
                   L.ptr(B["motion_id"]) + 4 * r0, L.ptr(B["motion_time"]) + 4 * r0, self._smp_c, self._num_clips,
                   L.ptr(Nm["d_sum"]) if record else None, st)
            plan.run(st)
            L.call("addhip_disc_reward", L.ptr(W["logits"]), L.ptr(B["reward"]) + 4 * r0, rows, self._disc_reward_scale, self._task_reward_weight,
                   self._disc_reward_weight, L.ptr(W["rstats"]), st)
        main.wait_event(joined)
        W["norm_diff"][:self.Mb + 1].zero_()  # row Mb must stay the zero-difference sample for the update plan
        L.call("addhip_sampler_update", self._smp_c, self._num_clips, st)
        succ = self._env.get_reward_succ() / (1.0 - self._discount)  # base_agent.py:472-480
        fail = self._env.get_reward_fail() / (1.0 - self._discount)
        L.call("addhip_td_lambda_adv", L.ptr(B["reward"]), L.ptr(B["vals"][1:]), L.ptr(B["timeout_vals"]), L.ptr(B["vals"]), L.ptr(B["done"]), L.ptr(B["rand_mask"]), T, N,
               self._discount, self._td_lambda, succ, fail, self._norm_adv_clip, L.ptr(B["tar_val"]), L.ptr(B["adv"]), L.ptr(W["scratch"]),
               L.ptr(W["adv_stats"]), st)

    def _next_minibatch_indices(self):
        """ExperienceBuffer._sample_rand_idx (experience_buffer.py:92-113): consecutive slices of a permutation of the
        T*N samples, redrawn when exhausted; device-side randperm (torch RNG is plumbing here)."""
        total, n = self.T * self.N, self.Mb
        if self.inject is not None and "perms" in self.inject:
            randperm = lambda: next(self.inject["perms"]).to(self._device)
        else:
            randperm = lambda: torch.randperm(total, device=self._device)
        if not hasattr(self, "_perm") or self._perm is None:
            self._perm, self._perm_head = randperm(), 0
        sample_count = min(self._total_samples, total)
        if self._perm_head + n <= total:
            idx = self._perm[self._perm_head:self._perm_head + n]
            self._perm_head += n
            if sample_count == total:  # the usual case (full buffer, slice inside one permutation): the gather reads the slice in place
                self._gather_c.idx = idx.data_ptr()
                return
        else:
            idx0 = self._perm[self._perm_head:]
            rem = n - (total - self._perm_head)
            self._perm = randperm()
            idx = torch.cat([idx0, self._perm[:rem]])
            self._perm_head = rem
        self._W["perm_idx"].copy_(torch.remainder(idx, sample_count))
        self._gather_c.idx = L.ptr(self._W["perm_idx"])

    def _update_model(self):
        """ppo_agent.py:171-192."""
        W, m = self._W, self._model
        st = self._stream()
        total = self.T * self.N
        num_batches = int(np.ceil(float(min(self._total_samples, total)) / self.Mb))
        W["stats"].zero_()
        # (the bias-gradient replica rows are zero between steps by construction -- every combine clears what its dX GEMM wrote; an update
        # phase still starts from a known state, e.g. after a tool replayed single launches of the plan)
        for r in (self._run_actor, self._run_critic, self._run_disc):
            r.bias_rep.zero_()
        steps, n_steps = 0, self._update_epochs * num_batches
        main, side = torch.cuda.current_stream(), self._side_streams[0]
        self._next_minibatch_indices()
        self._gather(st)
        while steps < n_steps:
            self._run_update_sections(zero_grad=steps == 0)  # later steps find the gradient zeroed by the optimiser launch before them
            self._post_exchange_grads(st)
            steps += 1
            gathered = None
            if steps < n_steps:
                # the next minibatch is gathered on a side stream beside this step's optimiser (the sections have joined: the
                # minibatch buffers are free; the gather reads nothing the optimiser writes)
                self._next_minibatch_indices()
                joined = torch.cuda.Event()
                joined.record(main)
                side.wait_event(joined)
                self._gather(side.cuda_stream)
                gathered = torch.cuda.Event()
                gathered.record(side)
            if self._grad_clip > 0.0:
                L.call("addhip_grad_clip", L.ptr(m.grads), m.count, self._grad_clip, L.ptr(W["scratch"]), None, st)
            m.opt_step += 1
            # MPOptimizer.step: the update, the bf16 shadow of the new parameters (bf16-storage mode; the transposed copies follow inside
            # the next step's sections) and the zero_grad of the next step in ONE launch
            o = self._opt_c
            o.step = m.opt_step
            L.call("addhip_optimizer_step", o, st)
            m.refresh_w_amax(st)  # (f16x2: the new parameters' tracked maximum, the scale of the next step's weight operands)
            m.refresh_dist(st)    # (actor_std_type CONSTANT: std and the log-probability constant of the new log-std)
            if gathered is not None:
                main.wait_event(gathered)
        return steps

    def _gather(self, stream):
        """ExperienceBuffer.sample of one minibatch (addhip_gather_minibatch); f16x2: the rows' tracked maxima start from zero."""
        if self._f16x2:
            L.call("addhip_fill_zero", L.ptr(self._W["mb_amax"]), 2 * L.AMAX_SLOTS, stream)
        L.call("addhip_gather_minibatch", self._gather_c, stream)

    def _run_update_sections(self, zero_grad=True):
        """The actor, critic and discriminator sections of the update plan are independent until the optimiser: they run on
        three streams, so that one net's tile write-out bursts and partial last waves of workgroups overlap another
        net's MFMA phases (all tiles of one GEMM launch are in phase with each other).  With more than one rank each
        section is followed, on its own stream, by the asynchronous all-reduce of that net's gradient bucket (the exchange
        step: mean gradient over ranks, RCCL over xGMI), which therefore also overlaps the other nets' GEMMs."""
        m = self._model
        main = torch.cuda.current_stream()
        streams = [main] + self._side_streams
        if zero_grad:  # MPOptimizer.step's zero_grad (mp_optimizer.py:14-16); inside an update phase addhip_optimizer_step has done it
            L.call("addhip_fill_zero", L.ptr(m.grads), m.count, main.cuda_stream)
        # (bf16-storage mode: the gather wrote the bf16 copies of the minibatch rows too; row Mb of norm_diff16 stays 0 from its allocation)
        # stream-ordered asynchronous collectives are an nccl (RCCL) property; any other backend (gloo in rehearsals) gets one
        # blocking all-reduce of the whole gradient after the join
        # (a 1-rank process group takes the same path: the collectives are identities there, which is how the single-GPU
        # tests exercise the issue order, the bucket ranges and the stream joins)
        exchange = self._distributed
        overlap = exchange and torch.distributed.get_backend() == "nccl"
        pending = []

        def on_bucket(bucket, stream_index):  # called by addhip_schedule_run where the bucket is final in the issue order of its stream
            with torch.cuda.stream(streams[stream_index]):
                pending.append(D.all_reduce_sum_async(m.grads[bucket[0]:bucket[1]]))

        # the sections on their streams: fork from `main`, event dependencies between sections, join back into `main`
        self._schedule.run([s.cuda_stream for s in streams], on_bucket if overlap else None)
        # (every loss coefficient of the plan carries 1/world, so the SUM over ranks is already the mean: no scaling pass)
        if overlap:  # (four buckets: actor tail, critic tail, the two first layers, discriminator -- each issued where it became final)
            D.wait_all(pending)
        elif exchange:
            D.all_reduce_sum_(m.grads)

    def _post_exchange_grads(self, stream):
        """Loss terms whose gradient is identical on every rank, added once BEHIND the exchange: the entropy bonus of a trainable log-std
        (ppo_agent.py:262-266 with distribution_gaussian_diag.py:96-99: d(-w * mean entropy) / d logstd_j = -w)."""
        m = self._model
        if m.dist is not None and self._action_entropy_weight != 0:
            L.call("addhip_l2_grad", L.ptr(m.logstd_ones), m.g("actor", "logstd"), L.NUM_DOF, -self._action_entropy_weight, None, stream)

    def _need_normalizer_update(self):
        return self._sample_count < self._normalizer_samples

    def _update_normalizers(self):
        """amp_agent.py:61-63 -> Normalizer.update (normalizer.py:37-80, all-reduced) + DiffNormalizer.update."""
        B, Nm, tk, st = self._B, self._Nrm, self._task, self._stream()
        rows = self.T * self.N
        L.call("addhip_norm_accum", L.ptr(B["obs"]), rows, tk.obs_stride, tk.obs_stride, L.ptr(Nm["obs_sum"]), L.ptr(Nm["obs_sumsq"]), st)
        count = rows
        if self._world > 1:
            count = rows * D.all_reduce_sum_(Nm["obs_sum"], Nm["obs_sumsq"])
        L.call("addhip_norm_merge", L.ptr(Nm["obs_mean"]), L.ptr(Nm["obs_std"]), L.ptr(Nm["obs_msq"]), L.ptr(Nm["obs_cnt"]), L.ptr(Nm["obs_sum"]),
               L.ptr(Nm["obs_sumsq"]), count, tk.obs_stride, 1e-8, int(self._obs_norm_first), st)
        self._obs_norm_first = False
        # pad columns of the obs rows are identically 0 -> mean 0, var clamps to min_var; keep their std at 1
        Nm["obs_std"][tk.obs_dim:] = 1.0
        L.call("addhip_diffnorm_merge", L.ptr(Nm["d_abs"]), L.ptr(Nm["d_cnt"]), L.ptr(Nm["d_sum"]), rows, tk.disc_stride, st)

    def _train_iter(self):
        """base_agent.py:353-374.  Returns the info dict with the reference's keys."""
        t0 = time.perf_counter()
        B = self._B
        # carry: the last obs of the previous iteration is the first of this one
        if self._iter_started:
            B["obs"][0].copy_(B["obs"][self.T])
        self._iter_started = True
        self._rollout_train()
        self._build_train_data()
        steps = self._update_model()
        if self._need_normalizer_update():
            self._update_normalizers()
        L.call("addhip_return_tracker_fold", L.ptr(B["ep_stats"]), self.T, L.ptr(self._train_state), self._stream())
        info = self._collect_info(steps)
        self._timers["iter_s"] = time.perf_counter() - t0
        return info

    def _collect_info(self, steps):
        """One device->host read per iteration: the 17 logged scalars (ppo_agent.py:190-192, add_agent.py:190-199)."""
        W = self._W
        s = W["stats"].double().cpu().numpy() / max(steps, 1)
        adv = W["adv_stats"].cpu().numpy()
        rs = W["rstats"].double().cpu().numpy()
        trk = self._train_state.cpu().numpy()
        Mb, M1 = float(self.Mb), float(self.Mb)
        actor_min, clipf, ratio, bound = -s[0], s[1], s[2], s[3]  # already per-minibatch means over the exploring samples
        actor_loss = actor_min + self._action_bound_weight * bound
        extra = {}
        if self._action_entropy_weight != 0:  # fixed-std policy: the entropy is a constant (distribution_gaussian_diag.py:96-99)
            # (trainable log-std: the entropy of the policy as it stands after the iteration's last step, not the mean over its steps)
            ent = self._model.entropy if self._model.dist is None else float(self._model.dist[33])
            if self._model.std_type == "VARIABLE":  # (per-sample entropies: their mean over the exploring samples, averaged over the steps)
                ent = s[6]
            actor_loss += -self._action_entropy_weight * ent
            extra["action_entropy"] = ent
        if self._action_reg_weight != 0:
            actor_loss += self._action_reg_weight * s[5]
            extra["action_reg_loss"] = s[5]
        critic_loss = s[8] / Mb
        bce_neg, bce_pos = s[12] / M1, s[13]
        gp = s[20] / Mb
        w_all, w_logit = s[24] + s[25], s[25]
        disc_loss = 0.5 * (bce_pos + bce_neg) + self._disc_logit_reg * w_logit + self._disc_grad_penalty * gp + self._disc_weight_decay * w_all
        n = self.T * self.N
        dr_mean = rs[0] / n
        dr_std = math.sqrt(max(rs[1] - n * dr_mean * dr_mean, 0.0) / max(n - 1, 1))
        info = {
            "loss": actor_loss + self._critic_loss_weight * critic_loss + self._disc_loss_weight * disc_loss,
            "critic_loss": critic_loss, "actor_loss": actor_loss, "clip_frac": clipf, "imp_ratio": ratio, "action_bound_loss": bound,
            "disc_loss": disc_loss, "disc_grad_penalty": gp, "disc_logit_loss": w_logit, "disc_pos_acc": s[17], "disc_neg_acc": s[16] / M1,
            "disc_pos_logit": s[15], "disc_neg_logit": s[14] / M1, "adv_mean": float(adv[0]), "adv_std": float(adv[1]),
            "disc_reward_mean": dr_mean, "disc_reward_std": dr_std,
            "mean_return": float(trk[1]), "mean_ep_len": float(trk[2]), "num_eps": int(trk[0]),
        }
        info.update(extra)
        return info

    def _init_train(self):
        if not self._is_restored:
            self._iter, self._sample_count = 0, 0
        self._total_samples = 0
        self._perm = None
        self._iter_started = False
        self._train_state.zero_()
        self._S["ret_acc"].zero_()
        self._S["len_acc"].zero_()

    def reset_all_envs(self, stream_id=None):
        B = self._B
        self._reset_envs(True, B["obs"][0], B["disc_obs"][self.T], B["disc_demo"][self.T], self.stream_reset_all(0) if stream_id is None else stream_id)
        # rows of envs that never ran into the time limit are evaluated by the critic too (and ignored): keep them ordinary
        # observations instead of zeros, which normalise to huge inputs wherever an obs column is nearly constant
        B["obs_timeout"].copy_(B["obs"][0])
        self._iter_started = False

    def train_model(self, out_model_file, int_output_dir, log_file):
        """base_agent.py:79-114."""
        start = time.time()
        self.reset_all_envs()
        self._logger = self._build_logger(log_file)
        self._init_train()
        test_info = {"mean_return": 0.0, "mean_ep_len": 0.0, "num_eps": 0}
        # everything built so far is long-lived: take it out of the cyclic GC's view so a full collection (tens of ms with
        # torch's object graph) cannot land between two kernel launches of the loop
        gc.collect()
        gc.freeze()
        while self._sample_count < self._max_samples:
            output_iter = self._iter % self._iters_per_output == 0
            if output_iter:
                test_info = self.test_model(self._test_episodes)
            train_info = self._train_iter()
            self._sample_count = self._total_samples
            if self._sample_count >= self._max_samples:
                output_iter = True
                test_info = self.test_model(self._test_episodes)
            self._log_train_info(train_info, test_info, start)
            self._logger.print_log()
            if output_iter:
                self._logger.write_log()
                self._output_train_model(self._iter, out_model_file, int_output_dir)
                self._train_state.zero_()
                self._S["ret_acc"].zero_()
                self._S["len_acc"].zero_()
                self.reset_all_envs(self.stream_reset_all(self._iter + 1))
            self._iter += 1

    def test_model(self, num_episodes):
        """base_agent.py:116-126, 393-425: deterministic (mode) actions until every env has finished its quota of episodes.
        (Host-synchronous by design: it is outside the throughput path.)

        task.reference_compat (default on) reproduces what the reference does here: Environment.set_mode(TEST) sets
        env.num_envs = 1 (envs/env.py:142-148), so the reset at the start touches env 0 only -- the other envs carry on from
        wherever training left them -- and the quota is ceil(num_episodes / 1) = num_episodes finished episodes for EVERY env.
        Corrected mode: all envs start from a fresh reset and the quota is ceil(num_episodes / num_envs) per env."""
        self.set_mode(AgentMode.TEST)
        if int(num_episodes) == 0:
            self.set_mode(AgentMode.TRAIN)
            return {"mean_return": 0.0, "mean_ep_len": 0.0, "num_eps": 0}
        B, S = self._B, self._S
        compat = bool(self._task_cfg.get("reference_compat", True))
        call = self._test_calls
        self._test_calls += 1
        if compat:
            S["done"].zero_()
            S["done"][0] = L.DONE_FAIL  # any non-NULL flag: the masked reset then re-samples exactly env 0
            self._reset_envs(False, B["obs"][0], B["disc_obs"][self.T], B["disc_demo"][self.T], self.stream_test_reset_all(call))
            min_eps = int(num_episodes)
        else:
            self.reset_all_envs(self.stream_test_reset_all(call))
            min_eps = int(np.ceil(num_episodes / self.N))
        S["ret_acc_test"].zero_()
        S["len_acc_test"].zero_()
        self._test_state.zero_()
        eps_per_env = torch.zeros(self.N, dtype=torch.long, device=self._device)
        ep = torch.zeros(1, 3, device=self._device)
        out = L.StepOutT(L.ptr(B["obs"][0]), None, None, L.ptr(B["disc_obs"][self.T]), L.ptr(B["disc_demo"][self.T]), None, None, None, None, L.ptr(ep))
        k = 0
        while True:
            self._decide_action(0, 0, True)
            if self.inject is not None and "pre_step" in self.inject:
                self.inject["pre_step"](k)
            ep.zero_()
            self._step_env(0, out, self._env_c_test)
            eps_per_env += (S["done"] != 0).long()
            L.call("addhip_return_tracker_fold", L.ptr(ep), 1, L.ptr(self._test_state), self._stream())
            self._reset_envs(False, B["obs"][0], B["disc_obs"][self.T], B["disc_demo"][self.T], self.stream_test_reset(call, k))
            k += 1
            if bool(torch.all(eps_per_env > min_eps - 1)):
                break
        st = self._test_state.cpu().numpy()
        self.set_mode(AgentMode.TRAIN)
        self._iter_started = False
        self._test_steps = k
        return {"mean_return": float(st[1]), "mean_ep_len": float(st[2]), "num_eps": int(st[0])}

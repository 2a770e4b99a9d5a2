"""Agent-level golden fixtures (task layer, learner, full iteration) taken from the reference's
ADDAgent running on tools/ref_harness.FakeEngine.  Build-container tooling only."""
import contextlib
import json
import os
import sys

import numpy as np
import torch

import ref_harness as H

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.learn import PARAM_SHAPES, param_shapes, synth_params  # noqa: E402  (shared deterministic weights)


def _save(name, **arrays):
    from gen_golden import save

    save(name, **arrays)


def T(x, dtype=torch.float32):
    return torch.tensor(np.asarray(x), dtype=dtype)


class DrawLog:
    """Records (and optionally replays) every torch RNG call the reference makes."""

    def __init__(self):
        self.calls = []

    @contextlib.contextmanager
    def recording(self):
        names = ("normal", "bernoulli", "multinomial", "rand", "randperm")
        orig = {n: getattr(torch, n) for n in names}

        def wrap(n):
            def f(*a, **k):
                out = orig[n](*a, **k)
                self.calls.append((n, out.clone()))
                return out

            return f

        for n in names:
            setattr(torch, n, wrap(n))
        try:
            yield self
        finally:
            for n in names:
                setattr(torch, n, orig[n])

    def take(self, name):
        out = [c for n, c in self.calls if n == name]
        return out


def build_agent(num_envs, clip_frames=200, seed=3, two_clip=False, task_over=None, model_over=None, **agent_over):
    from add_gym.learning.add.add_agent import ADDAgent

    if two_clip:
        from gen_golden import two_clip_yaml

        mf = two_clip_yaml()[0]
    else:
        mf = H.scratch_clip("walk1_subject1_trimmed.motion", clip_frames)
    cfg = H.load_ref_config(num_envs, mf, **agent_over)
    if task_over:
        cfg["task"].update(task_over)
    if model_over:
        cfg["agent"]["model"].update(model_over)
    torch.manual_seed(seed)
    ag = ADDAgent(cfg)
    return ag, cfg


def load_synth(ag, seed, nets=None, logstd=False, disc_dim=114):
    sd = ag.state_dict()
    for k, v in synth_params(seed, nets=nets, logstd=logstd, disc_dim=disc_dim).items():
        assert tuple(sd[k].shape) == v.shape, k
        sd[k] = T(v)
    ag.load_state_dict(sd)


def randomize_sim(ag, rng, lib_time_max, at_times=None):
    """Plausible simulator state near the reference pose + noise; returns nothing (state lives
    in the fake entity)."""
    ent = ag._env.robot.entity
    n = ent.n
    obs = ag._add_obs
    ids = obs._motion_ids
    t = T(rng.rand(n).astype(np.float32) * lib_time_max)
    if at_times is not None:
        t = at_times
    rp, rr, rv, ra, dp, dv = ag._add_motion.get_motion_step(ids, t)
    noise = lambda s, shape: T(rng.standard_normal(shape).astype(np.float32) * s)
    q = rr + noise(0.05, (n, 4))
    q = q / q.norm(dim=-1, keepdim=True)
    ent.pos[:] = rp + noise(0.05, (n, 3))
    ent.quat[:] = q
    ent.vel[:] = rv + noise(0.2, (n, 3))
    ent.ang[:] = ra + noise(0.2, (n, 3))
    ent.dofs_pos[:, 6:] = dp + noise(0.1, (n, 29))
    ent.dofs_vel[:, 6:] = dv + noise(0.5, (n, 29))
    ent.dofs_vel[:, 0:3] = ent.vel
    ent.dofs_vel[:, 3:6] = ent.ang


def sim_arrays(ag):
    r = ag._env.robot
    return dict(root_pos=r.base_pos.clone(), root_rot=r.base_quat.clone(), root_vel=r.base_lin_vel.clone(),
                root_ang_vel=r.base_ang_vel.clone(), dof_pos=r.dof_pos.clone(), dof_vel=r.dof_vel.clone())


def hist_arrays(obs):
    names = ("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_pos", "dof_vel")
    out = {"hist_" + n: getattr(obs, "_disc_hist_" + n)._buffer.clone() for n in names}
    out["hist_head"] = obs._disc_hist_root_pos._head
    return out


# ------------------------------------------------------------------------------------
# joint_err_w (add_reward.py:24-52): one weight per joint in kinematic-tree order (all 29 G1 joints are 1-dof)
JOINT_ERR_W = [0.5, 0.5, 2.0, 1.0, 1.0, 2.0, 1.5, 1.5, 0.25, 3.0, 3.0, 1.0, 0.75, 0.75, 1.0, 0.1, 0.1, 1.0, 2.5, 2.5, 1.0, 1.0, 1.0, 0.3, 0.3, 0.3,
               0.3, 0.0, 0.0]


def gen_obs_reward_done_jw():
    """A separate fixture (the first one stays byte-identical): non-uniform joint error weights."""
    gen_obs_reward_done({"joint_w": dict(joint_err_w=JOINT_ERR_W)}, "obs_reward_done_jw")


def gen_obs_reward_done_s2(S=2, word="two"):
    """task.num_disc_obs_steps = 2 / 4 (add_observation.py:276-294, 362-375): an S-deep history ring, S clip frames per demo observation."""
    gen_obs_reward_done({f"{word}_steps": dict(num_disc_obs_steps=S), f"{word}_steps_local_vel": dict(num_disc_obs_steps=S, global_obs=False, enable_vel_obs=True)},
                        f"obs_reward_done_s{S}")


def gen_obs_reward_done(variants=None, name="obs_reward_done"):
    variants = variants or {
        "default": {},
        "local": dict(global_obs=False),
        "noheight": dict(root_height_obs=False),
        "local_noheight": dict(global_obs=False, root_height_obs=False),
        "vel_phase": dict(enable_vel_obs=True, enable_phase_obs=True),
        "local_vel": dict(global_obs=False, enable_vel_obs=True),
    }
    out = {}
    n = 32
    for vname, over in variants.items():
        ag, cfg = build_agent(n, task_over=over)
        rng = np.random.RandomState(21)
        ag._reset_envs()
        obs = ag._add_obs
        env = ag._env
        ent = env.robot.entity
        # a few warm-up steps so that the ring head is not 0 and histories differ per slot
        for _ in range(2):
            randomize_sim(ag, rng, 6.0)
            ag._step_env(torch.zeros(n, 29))
        # craft time/offset so that TIME, SUCC and FAIL all occur
        env.time_buf[:] = T(rng.rand(n).astype(np.float32) * 5.0)
        obs._motion_time_offsets[:] = T(np.floor(rng.rand(n) * 100).astype(np.float32) * 0.01 + 0.02)
        env.time_buf[0:4] = 19.99  # -> >= 20 after += dt : TIME (or SUCC since past the clip end)
        obs._motion_time_offsets[0:4] = 0.02
        obs._motion_time_offsets[0:2] = -15.0  # motion time 5.0 < clip length: TIME survives (flag-logic case)
        env.time_buf[4:8] = 6.5  # past the 6.633 s clip end -> SUCC
        obs._motion_time_offsets[4:8] = 0.2
        randomize_sim(ag, rng, 6.0, at_times=env.time_buf + obs._motion_time_offsets + 0.01)
        ent.dofs_pos[8:10, 6:] += 3.0  # pose failure
        ent.pos[10:12, 0] += 2.0  # root failure (only when tracking the global root)
        ent.forced_contact_link[:] = -1
        ent.forced_contact_link[12] = 5  # non-foot link (left_hip_roll? any non-contact body) -> FAIL
        ent.forced_contact_link[13] = 19  # left_ankle_roll_link: allowed contact -> no FAIL
        pre = dict(time=env.time_buf.clone(), time_off=obs._motion_time_offsets.clone(), motion_ids=obs._motion_ids.clone())
        pre.update(hist_arrays(obs))
        target = T(rng.standard_normal((n, 29)).astype(np.float32) * 0.3)
        pre["sim_pre_dof_pos"] = env.robot.dof_pos.clone()
        o, r, d, info = ag._step_env(target)
        post = sim_arrays(ag)
        contact = env.robot.get_ground_contact_forces_v2(env.plane, ag._add_done._noncontact_body_ids)
        res = dict(action=target, obs=o.clone(), reward=r.clone(), done=d.clone(), disc_obs=info["disc_obs"].clone(),
                   disc_obs_demo=info["disc_obs_demo"].clone(), contact=contact, time_post=env.time_buf.clone())
        res.update({"ref_" + k: getattr(obs, "ref_" + k).clone() for k in ("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_pos", "dof_vel")})
        for k, v in {**pre, **post, **res}.items():
            out[f"{vname}.{k}"] = v
        out[f"{vname}.noncontact_ids"] = ag._add_done._noncontact_body_ids
        if "joint_err_w" in over:
            out[f"{vname}.dof_err_w"] = ag._add_reward._dof_err_w.clone()
    _save(name, **out)


def gen_reset_s2(S=2):
    gen_reset(f"reset_s{S}", dict(num_disc_obs_steps=S), (("one", False),))


def gen_reset(name="reset", task_over=None, tags=(("one", False), ("two", True))):
    n = 48
    out = {}
    for tag, two in tags:
        ag, cfg = build_agent(n, two_clip=two, task_over=task_over)
        rng = np.random.RandomState(4)
        ag._reset_envs()
        for _ in range(2 if not two else 1):
            randomize_sim(ag, rng, 1.0)
            ag._step_env(torch.zeros(n, 29))
        # non-trivial sampler errors
        ag._add_motion.sampler.errors = T(rng.rand(*ag._add_motion.sampler.errors.shape).astype(np.float32) * 3 + 0.1)
        env_ids = T(np.sort(rng.choice(n, 17, replace=False)), torch.long)
        pre = hist_arrays(ag._add_obs)
        pre.update(time=ag._env.time_buf.clone(), time_off=ag._add_obs._motion_time_offsets.clone(),
                   motion_ids=ag._add_obs._motion_ids.clone(), sampler_errors=ag._add_motion.sampler.errors.clone())
        pre.update({"sim_" + k: v for k, v in sim_arrays(ag).items()})
        log = DrawLog()
        with log.recording():
            o, info = ag._reset_envs(env_ids)
        mult = log.take("multinomial")
        post = {"post_" + k: v for k, v in hist_arrays(ag._add_obs).items()}
        post.update({"post_sim_" + k: v for k, v in sim_arrays(ag).items()})
        res = dict(env_ids=env_ids, draw_ids=mult[0], draw_segments=mult[1].squeeze(-1), draw_jitter=log.take("rand")[0],
                   post_time=ag._env.time_buf.clone(), post_time_off=ag._add_obs._motion_time_offsets.clone(),
                   post_motion_ids=ag._add_obs._motion_ids.clone(), obs=o.clone(), disc_obs=info["disc_obs"].clone(),
                   disc_obs_demo=info["disc_obs_demo"].clone(), probs=ag._add_motion.sampler.get_probs(mult[0]))
        for k, v in {**pre, **post, **res}.items():
            out[f"{tag}.{k}"] = v
    _save(name, **out)


def gen_sampler():
    from add_gym.learning.sampler import AdaptiveSegmentSampler

    rng = np.random.RandomState(9)
    lengths = T([6.6333, 2.0, 9.3])
    s = AdaptiveSegmentSampler(lengths, 0.01, 20, None, 0.02)
    m = 4000
    ids = T(rng.randint(0, 3, m), torch.long)
    ids[:50] = 1
    times = T(rng.rand(m).astype(np.float32)) * lengths[ids] * 1.05
    err = T(rng.rand(m).astype(np.float32) * 4)
    s.update_errors(ids, times, err)
    e1 = s.errors.clone()
    ids2 = T(rng.randint(0, 2, 300), torch.long)  # clip 2 untouched this round
    times2 = T(rng.rand(300).astype(np.float32)) * lengths[ids2]
    err2 = T(rng.rand(300).astype(np.float32))
    s.update_errors(ids2, times2, err2)
    q = T([0, 2, 2, 1], torch.long)
    # floor-divide quantisation on its own
    tt = T(rng.rand(2000).astype(np.float32) * 10)
    _save("sampler", lengths=lengths, ids=ids, times=times, err=err, errors1=e1, ids2=ids2, times2=times2, err2=err2,
          errors2=s.errors, probs_ids=q, probs=s.get_probs(q), probs_all=s.get_probs(), segment_sizes=s.segment_sizes,
          fd_in=tt, fd_out=(tt // 0.01) * 0.01)


def gen_actor_step(name="actor_step", logstd=False):
    n = 96
    ag, cfg = build_agent(n, model_over={"actor_std_type": "VARIABLE" if logstd == "variable" else "CONSTANT"} if logstd else None)
    load_synth(ag, 101, logstd=logstd)
    rng = np.random.RandomState(2)
    ag._reset_envs()
    randomize_sim(ag, rng, 6.0)
    obs, _ = ag._reset_envs(T([], torch.long))
    obs = ag._add_obs._compute_obs()
    # non-trivial obs normaliser
    ag._obs_norm.record(obs + T(rng.standard_normal(obs.shape).astype(np.float32)))
    ag._obs_norm.update()
    log = DrawLog()
    with log.recording(), torch.no_grad():
        a, a_info = ag._decide_action(obs, None)
    _save(name, seed=101, obs=obs, obs_mean=ag._obs_norm._mean, obs_std=ag._obs_norm._std,
          a_mean=ag._a_norm._mean, a_std=ag._a_norm._std, noise=log.take("normal")[0], action=a, a_logp=a_info["a_logp"],
          rand_action_mask=a_info["rand_action_mask"])


def gen_td_lambda_adv():
    import add_gym.learning.base_agent as ba

    n, Tn = 16, 32
    ag, cfg = build_agent(n)
    rng = np.random.RandomState(13)
    r = T(rng.rand(Tn, n).astype(np.float32) * 2)
    done = T(rng.choice([0, 0, 0, 0, 0, 0, 1, 2, 3], size=(Tn, n)), torch.int32)
    nv = T(rng.standard_normal((Tn, n)).astype(np.float32) * 3)
    v = T(rng.standard_normal((Tn, n)).astype(np.float32) * 3)
    ret = ba.compute_td_lambda_return(r, nv, done, 0.99, 0.95)
    # run the reference's own advantage code: feed crafted critic outputs through _build_train_data
    eb = ag._exp_buffer
    eb._buffers["reward"][:] = r
    eb._buffers["done"][:] = done
    eb._buffers["rand_action_mask"][:] = 1.0
    queue = [nv.unsqueeze(-1).clone(), v.unsqueeze(-1).clone()]
    ag.model.eval_critic = lambda obs: queue.pop(0)
    from add_gym.learning import ppo_agent

    info = ppo_agent.PPOAgent._build_train_data(ag)
    _save("td_lambda_adv", r=r, done=done, next_vals=nv, vals=v, ret_raw=ret, tar_val=eb.get_data("tar_val"),
          adv=eb.get_data("adv"), adv_mean=info["adv_mean"], adv_std=info["adv_std"])


def param_summary(named):
    out = {}
    for k, v in named.items():
        f = v.detach().reshape(-1).double()
        stride = max(1, f.numel() // 64)
        out[k + ".sum"] = f.sum()
        out[k + ".l2"] = f.square().sum().sqrt()
        out[k + ".sample"] = v.detach().reshape(-1)[::stride][:64].clone()
    return out


SMALL_NETS = dict(actor_net="fc_2layers_256units", critic_net="fc_2layers_512units", disc_net="fc_2layers_128units")


def gen_losses_small_nets():
    """The same minibatch through other modules of the reference's net registry (nets/net_builder.py:5-11): a two-layer actor and
    critic of different widths and a 128/64 discriminator."""
    gen_losses("losses_small_nets", SMALL_NETS)


def gen_losses_disc3():
    """A three-hidden-layer discriminator (fc_3layers_1024units: the only deeper module of the registry): the gradient penalty's double
    backward runs through one more layer."""
    gen_losses("losses_disc3", dict(disc_net="fc_3layers_1024units"))


def gen_losses_constant_std():
    """actor_std_type CONSTANT (distribution_gaussian_diag.py:32-37): the log-std is a trainable vector, different per action dimension here."""
    gen_losses("losses_constant_std", None, logstd=True)


def gen_losses_variable_std():
    """actor_std_type VARIABLE (distribution_gaussian_diag.py:38-43, 52-53): the log-std is a second linear head on the actor's last layer."""
    gen_losses("losses_variable_std", None, logstd="variable")


def gen_losses_variable_std_entropy():
    gen_losses("losses_variable_std_entropy", None, logstd="variable", agent_over=dict(action_entropy_weight=0.05))


def gen_losses_constant_std_entropy():
    """the same with the entropy bonus on (action_entropy_weight = 0.05, ppo_agent.py:262-266): its gradient reaches the log-std only."""
    gen_losses("losses_constant_std_entropy", None, logstd=True, agent_over=dict(action_entropy_weight=0.05))


def gen_losses(name="losses", nets=None, logstd=False, agent_over=None):
    n = 64
    M = 256
    ag, cfg = build_agent(n, model_over=dict(nets or {}, **({"actor_std_type": "VARIABLE" if logstd == "variable" else "CONSTANT"} if logstd else {})), **(agent_over or {}))
    load_synth(ag, 202, nets, logstd)
    rng = np.random.RandomState(17)
    # non-trivial normalisers
    ag._obs_norm._mean[:] = T(rng.standard_normal(264).astype(np.float32) * 0.3)
    ag._obs_norm._std[:] = T(rng.rand(264).astype(np.float32) + 0.5)
    ag._disc_obs_norm._mean_abs[:] = T(rng.rand(114).astype(np.float32) * 0.5 + 0.05)
    batch = dict(
        obs=T(rng.standard_normal((M, 264)).astype(np.float32)),
        action=ag._a_norm.unnormalize(T(rng.standard_normal((M, 29)).astype(np.float32) * 0.6)),
        a_logp=T(rng.standard_normal(M).astype(np.float32) * 3 + 40),
        adv=T(np.clip(rng.standard_normal(M), -4, 4).astype(np.float32)),
        tar_val=T(rng.standard_normal(M).astype(np.float32) * 2),
        rand_action_mask=T((rng.rand(M) < 0.9).astype(np.float32)),
        disc_obs=T(rng.standard_normal((M, 114)).astype(np.float32)),
        disc_obs_demo=T(rng.standard_normal((M, 114)).astype(np.float32)),
    )
    # make old log-probs consistent with the current policy so that ratios are O(1)
    with torch.no_grad():
        dist = ag.model.eval_actor(ag._obs_norm.normalize(batch["obs"]))
        if logstd == "variable":
            # actions drawn from the policy itself (a few sigma at most): with the 12-sigma actions above d logp / d logstd ~ 140 per dimension,
            # and ONE Adam step through the log-std head moves logp by ~1e2 -- the reference's own third step is NaN
            batch["action"] = ag._a_norm.unnormalize(dist.mean + dist.stddev * T(rng.standard_normal((M, 29)).astype(np.float32) * 1.5))
        batch["a_logp"] = dist.log_prob(ag._a_norm.normalize(batch["action"])) + T(rng.standard_normal(M).astype(np.float32) * 0.3)
    inp = {k: v.clone() for k, v in batch.items()}
    out = {"in." + k: v for k, v in inp.items()}
    out.update(obs_mean=ag._obs_norm._mean, obs_std=ag._obs_norm._std, disc_mean_abs=ag._disc_obs_norm._mean_abs,
               a_mean=ag._a_norm._mean, a_std=ag._a_norm._std, seed=202)
    names = list(synth_params(202, nets=nets, logstd=logstd))  # (registration order)
    assert names == [n_ for n_, p_ in ag.named_parameters() if p_.requires_grad and n_ in names] and len(names) == sum(p_.requires_grad for p_ in ag.parameters())
    if nets is not None:
        out["nets"] = np.array(json.dumps(nets))
    if logstd:
        out["logstd"] = np.array(2 if logstd == "variable" else 1)
    if agent_over:
        out["agent_over"] = np.array(json.dumps(agent_over))
    sd_params = dict(ag.named_parameters())
    for step in range(3):
        info = ag._compute_loss({k: v.clone() for k, v in inp.items()})
        ag._optimizer.step(info["loss"])
        if step == 0:
            for k, v in info.items():
                out["info." + k] = v.detach()
            out.update({"grad." + k: v for k, v in param_summary({n_: sd_params[n_].grad for n_ in names}).items()})
        if step in (0, 2):
            out.update({f"param{step + 1}." + k: v for k, v in param_summary({n_: sd_params[n_] for n_ in names}).items()})
    _save(name, **out)


def gen_normalizers():
    from add_gym.learning.normalizer import Normalizer
    from add_gym.learning.diff_normalizer import DiffNormalizer

    rng = np.random.RandomState(31)
    nm = Normalizer((7,), "cpu")
    dn = DiffNormalizer((5,), "cpu")
    out = {}
    for it in range(3):
        for s in range(4):
            x = T(rng.standard_normal((33, 7)).astype(np.float32) * (1 + it) + it)
            y = T(rng.standard_normal((33, 5)).astype(np.float32) * 0.01 * (1 + it))
            y[:, 0] *= 1e-4
            nm.record(x)
            dn.record(y)
            out[f"x{it}_{s}"], out[f"y{it}_{s}"] = x, y
        nm.update()
        dn.update()
        out[f"mean{it}"], out[f"std{it}"], out[f"count{it}"] = nm._mean.clone(), nm._std.clone(), nm._count.clone()
        out[f"mean_abs{it}"], out[f"dcount{it}"] = dn._mean_abs.clone(), dn._count.clone()
    xq = T(rng.standard_normal((9, 7)).astype(np.float32))
    yq = T(rng.standard_normal((9, 5)).astype(np.float32))
    out.update(xq=xq, yq=yq, xq_norm=nm.normalize(xq), xq_unnorm=nm.unnormalize(xq), yq_norm=dn.normalize(yq))
    _save("normalizers", **out)


def gen_loop_1iter_two():
    """The whole iteration on the TWO-clip library (weights 1:3): clip ids are drawn per reset, the sampler table is [2,20], and the
    step tables are indexed with the raw-frame clip offsets (motion_lib.py:280-282, 322-326)."""
    gen_loop_1iter(name="loop_1iter_two", two_clip=True)


def gen_loop_1iter_time():
    """The whole iteration with a 0.4 s episode limit and envs that are already up to 0.36 s into their episodes, so that DONE_TIME
    samples occur: the critic then bootstraps from V(next_obs) of the PRE-reset observation (ppo_agent.py:117-133)."""
    gen_loop_1iter(name="loop_1iter_time", task_over={"max_episode_length": 0.4}, time_preset=True)


def gen_loop_1iter(name="loop_1iter", two_clip=False, task_over=None, time_preset=False):
    n = 32
    ag, cfg = build_agent(n, seed=5, two_clip=two_clip, task_over=task_over)
    load_synth(ag, 303, disc_dim=38 * int((task_over or {}).get("num_disc_obs_steps", 3)))
    log0 = DrawLog()
    with log0.recording():
        ag._curr_obs, ag._curr_info = ag._reset_envs()
    preset = torch.zeros(n)
    if time_preset:
        # shift the episode clock without moving the motion clock: k/64 is exact in fp32 and so is (offset - k/64)
        preset = T(np.random.RandomState(12).randint(0, 24, n).astype(np.float32) / 64.0)
        ag._env.time_buf[:] = ag._env.time_buf + preset
        ag._add_obs._motion_time_offsets[:] = ag._add_obs._motion_time_offsets - preset
    ag._exp_buffer.clear()
    perm0 = ag._exp_buffer._sample_buf.clone()
    # force some early terminations through fake contacts at chosen steps
    ent = ag._env.robot.entity
    rng = np.random.RandomState(8)
    Tn = cfg["agent"]["steps_per_iter"]
    contact_plan = np.full((Tn, n), -1, np.int64)
    for t in (3, 9, 10, 20, 27):
        contact_plan[t, rng.choice(n, 3, replace=False)] = 5
    step_counter = {"t": 0}
    orig_step = ag._step_env

    def step_env(action):
        ent.forced_contact_link[:] = T(contact_plan[step_counter["t"]], torch.long)
        step_counter["t"] += 1
        return orig_step(action)

    ag._step_env = step_env
    log = DrawLog()
    with log.recording():
        info = ag._train_iter()
    # split the draw log per step: each step = normal, bernoulli, [multinomial, multinomial, rand]
    noise, resets = [], []
    calls = log.calls
    i = 0
    for t in range(Tn):
        assert calls[i][0] == "normal" and calls[i + 1][0] == "bernoulli"
        noise.append(calls[i][1])
        i += 2
        if calls[i][0] == "multinomial":
            resets.append((calls[i][1], calls[i + 1][1].squeeze(-1), calls[i + 2][1]))
            i += 3
        else:
            resets.append((torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), torch.zeros(0)))
    perms = [perm0] + [c for nme, c in calls[i:] if nme == "randperm"]
    kmax = max(len(r[0]) for r in resets)
    pad = lambda x, fill, dt: torch.cat([x.to(dt), torch.full((kmax - len(x),), fill, dtype=dt)])
    eb = ag._exp_buffer
    out = dict(
        seed=303, contact_plan=contact_plan, noise=torch.stack(noise), time_preset=preset,
        reset_count=np.asarray([len(r[0]) for r in resets]),
        reset_ids=torch.stack([pad(r[0], 0, torch.long) for r in resets]),
        reset_segments=torch.stack([pad(r[1], 0, torch.long) for r in resets]),
        reset_jitter=torch.stack([pad(r[2], 0.0, torch.float32) for r in resets]),
        init_ids=log0.take("multinomial")[0], init_segments=log0.take("multinomial")[1].squeeze(-1), init_jitter=log0.take("rand")[0],
        perms=torch.stack(perms),
        obs_mean=ag._obs_norm._mean, obs_std=ag._obs_norm._std, disc_mean_abs=ag._disc_obs_norm._mean_abs,
        sampler_errors=ag._add_motion.sampler.errors,
    )
    for k in ("obs", "next_obs", "action", "reward", "done", "a_logp", "tar_val", "adv", "disc_obs", "disc_obs_demo", "motion_times"):
        v = eb.get_data(k)
        out["buf." + k + ".sum"] = v.double().sum()
        out["buf." + k + ".abs"] = v.double().abs().sum()
    out["buf.done"] = eb.get_data("done").clone()
    out["buf.reward"] = eb.get_data("reward").clone()
    out["buf.adv"] = eb.get_data("adv").clone()
    out["buf.motion_times"] = eb.get_data("motion_times").clone()
    out["buf.obs_last"] = eb.get_data("obs")[-1].clone()
    for k, v in info.items():
        out["info." + k] = float(v)
    names = [n_ for n_, _ in PARAM_SHAPES]
    sd_params = dict(ag.named_parameters())
    out.update({"param." + k: v for k, v in param_summary({n_: sd_params[n_] for n_ in names}).items()})
    _save(name, **out)


def gen_logger():
    """log.txt bytes and console lines of the reference's Logger / TBLogger (util/logger.py, util/tb_logger.py) for a fixed set of
    rows with the value types BaseAgent._log_train_info produces (ints for Iteration / Samples / *_Episodes, floats otherwise)."""
    import contextlib
    import io
    import json
    import tempfile

    from add_gym.util import tb_logger

    rows = [
        [("Iteration", 0, "1_Info", False), ("Wall_Time", 0.00012345678, "1_Info", False), ("Samples", 131072, "1_Info", False),
         ("Test_Return", 0.0, "0_Main", False), ("Test_Episode_Length", 0.0, "0_Main", True), ("Test_Episodes", 0, "1_Info", True),
         ("Train_Return", 12.3456789012, "0_Main", False), ("Train_Episode_Length", 33.25, "0_Main", True), ("Train_Episodes", 417, "1_Info", True),
         ("Loss", -0.00123456789, None, False), ("Clip_Frac", 1e-05, None, False), ("Disc_Pos_Acc", 1.0, None, False), ("Exp_Prob", 1.0, None, False)],
        [("Iteration", 100, "1_Info", False), ("Wall_Time", 1.5, "1_Info", False), ("Samples", 13238272, "1_Info", False),
         ("Test_Return", 123456.789, "0_Main", False), ("Test_Episode_Length", 1999.5, "0_Main", True), ("Test_Episodes", 4096, "1_Info", True),
         ("Train_Return", 1e-12, "0_Main", False), ("Train_Episode_Length", 2000.0, "0_Main", True), ("Train_Episodes", 1234567, "1_Info", True),
         ("Loss", 3.0e+20, None, False), ("Clip_Frac", 0.3333333333333333, None, False), ("Disc_Pos_Acc", 0.5, None, False), ("Exp_Prob", 0.2, None, False)],
    ]
    d = tempfile.mkdtemp(prefix="addgym_log_")
    path = os.path.join(d, "log.txt")
    lg = tb_logger.TBLogger()
    lg.set_step_key("Samples")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        lg.configure_output_file(path)
    console = []
    for row in rows:
        for key, val, col, quiet in row:
            lg.log(key, val, collection=col, quiet=quiet)
        b = io.StringIO()
        with contextlib.redirect_stdout(b):
            lg.print_log()
        console.append(b.getvalue())
        lg.write_log()
    lg.output_file.flush()
    with open(path, "rb") as f:
        raw = f.read()
    tags = lg._build_key_tags()
    _save("logger", rows=np.frombuffer(json.dumps(rows).encode(), np.uint8), log_txt=np.frombuffer(raw, np.uint8),
          console=np.frombuffer(json.dumps(console).encode(), np.uint8), tags=np.frombuffer(json.dumps(tags).encode(), np.uint8))


def gen_state_dict():
    """Names, shapes and dtypes of the reference agent's checkpoint: state_dict() (base_agent.py:148-155) and the optimizer's."""
    import json

    ag, cfg = build_agent(8, clip_frames=60)
    sd = ag.state_dict()
    model = [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()]
    ag._optimizer._optimizer.zero_grad()
    # one optimiser step so that the AdamW state exists
    loss = sum((p ** 2).sum() for p in ag._model.parameters() if p.requires_grad)
    ag._optimizer.step(loss)
    osd = ag._optimizer._optimizer.state_dict()
    opt_state = [[int(i), {k: (list(v.shape) if torch.is_tensor(v) else None) for k, v in st.items()}] for i, st in osd["state"].items()]
    group = {k: (v if not isinstance(v, (list, tuple)) or k == "betas" else len(v)) for k, v in osd["param_groups"][0].items()}
    import tempfile

    path = os.path.join(tempfile.mkdtemp(prefix="addgym_ck_"), "model.pt")
    ag.save(path)
    ck = torch.load(path, weights_only=True)
    top = sorted(ck.keys())
    blob = dict(model=model, opt_state=opt_state, opt_group=group, top=top, iter_type=type(ck["iter"]).__name__, sample_count_type=type(ck["sample_count"]).__name__)
    _save("state_dict", meta=np.frombuffer(json.dumps(blob, default=str).encode(), np.uint8))


def gen_test_rollout():
    """BaseAgent.test_model -> _rollout_test (base_agent.py:116-126, 393-425) on the fake engine, as train_model reaches it on an
    output iteration: reset of all envs (TRAIN mode), then test_model(n): deterministic (mode) actions, every reset draw
    recorded.  NB Environment.set_mode(TEST) sets env.num_envs = 1 (envs/env.py:142-148), so (i) the reset at the start of
    test_model touches env 0 only and (ii) the quota is ceil(n / 1) = n finished episodes for EVERY env."""
    n, episodes = 16, 3
    ag, cfg = build_agent(n, clip_frames=40, seed=9)  # 1.3 s clip: episodes end by SUCC after at most ~130 steps
    load_synth(ag, 404)
    log0 = DrawLog()
    with log0.recording():
        ag._curr_obs, ag._curr_info = ag._reset_envs()
    dones, rewards = [], []
    orig_step = ag._step_env

    def step_env(action):
        out = orig_step(action)
        rewards.append(out[1].clone())
        dones.append(out[2].clone())
        return out

    ag._step_env = step_env
    log = DrawLog()
    with log.recording():
        info = ag.test_model(episodes)
    calls = log.calls
    assert [c[0] for c in calls[:3]] == ["multinomial", "multinomial", "rand"] and len(calls[0][1]) == 1
    first = (calls[0][1], calls[1][1].squeeze(-1), calls[2][1])
    steps = len(dones)
    resets, i = [], 3
    for t in range(steps):
        k = int((dones[t] != 0).sum())
        if k > 0:
            assert calls[i][0] == "multinomial" and len(calls[i][1]) == k
            resets.append((calls[i][1], calls[i + 1][1].squeeze(-1), calls[i + 2][1]))
            i += 3
        else:
            resets.append((torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long), torch.zeros(0)))
    assert i == len(calls), (i, len(calls))
    kmax = max(len(r[0]) for r in resets)
    pad = lambda x, fill, dt: torch.cat([x.to(dt), torch.full((kmax - len(x),), fill, dtype=dt)])
    _save("test_rollout", seed=404, num_envs=n, episodes=episodes, clip_frames=40, steps=steps,
          init_ids=log0.take("multinomial")[0], init_segments=log0.take("multinomial")[1].squeeze(-1), init_jitter=log0.take("rand")[0],
          first_ids=first[0], first_segments=first[1], first_jitter=first[2],
          reset_count=np.asarray([len(r[0]) for r in resets]),
          reset_ids=torch.stack([pad(r[0], 0, torch.long) for r in resets]),
          reset_segments=torch.stack([pad(r[1], 0, torch.long) for r in resets]),
          reset_jitter=torch.stack([pad(r[2], 0.0, torch.float32) for r in resets]),
          done=torch.stack(dones), reward=torch.stack(rewards),
          mean_return=float(info["mean_return"]), mean_ep_len=float(info["mean_ep_len"]), num_eps=int(info["num_eps"]))


AGENT_GENS = dict(obs_reward_done_s4=lambda: gen_obs_reward_done_s2(4, "four"), reset_s4=lambda: gen_reset_s2(4),
                  loop_1iter_s4=lambda: gen_loop_1iter("loop_1iter_s4", task_over=dict(num_disc_obs_steps=4)), obs_reward_done_s2=gen_obs_reward_done_s2, reset_s2=gen_reset_s2, loop_1iter_s2=lambda: gen_loop_1iter("loop_1iter_s2", task_over=dict(num_disc_obs_steps=2)), loop_1iter_two=gen_loop_1iter_two, loop_1iter_time=gen_loop_1iter_time, logger=gen_logger, state_dict=gen_state_dict, test_rollout=gen_test_rollout, obs_reward_done=gen_obs_reward_done, obs_reward_done_jw=gen_obs_reward_done_jw, reset=gen_reset, sampler=gen_sampler, actor_step=gen_actor_step,
                  td_lambda_adv=gen_td_lambda_adv, losses=gen_losses, losses_small_nets=gen_losses_small_nets, losses_disc3=gen_losses_disc3, losses_variable_std=gen_losses_variable_std, losses_variable_std_entropy=gen_losses_variable_std_entropy, actor_step_variable_std=lambda: gen_actor_step("actor_step_variable_std", "variable"), losses_constant_std_entropy=gen_losses_constant_std_entropy, losses_constant_std=gen_losses_constant_std, actor_step_constant_std=lambda: gen_actor_step("actor_step_constant_std", True), normalizers=gen_normalizers, loop_1iter=gen_loop_1iter)

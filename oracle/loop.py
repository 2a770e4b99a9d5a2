"""One training iteration (rollout + build-train-data + update) of the ADD/PPO agent on the
CPU, with every random draw supplied by the caller (oracle restatement of
BaseAgent._train_iter and friends: base_agent.py:353-391, ppo_agent.py:111-192,
add_agent.py:93-233).  Test infrastructure; also the `cpu_baseline` leg of bench.py."""
import numpy as np
import torch

from . import learn as L
from . import task as T

F = np.float32


class KinematicSim:
    """Stand-in simulator behind the engine boundary (NOT a reference component: Genesis is
    out of scope).  Joints move half-way to their PD target each step, the root keeps the
    state written at reset.  Same model as tools/ref_harness.FakeEntity and the product's
    KinematicEngine, so that all three see identical simulator states."""

    LAG = F(0.5)

    def __init__(self, n, num_dof, dt):
        self.dt = dt
        self.root_pos = np.zeros((n, 3), F)
        self.root_pos[:, 2] = 0.793
        self.root_rot = np.zeros((n, 4), F)
        self.root_rot[:, 0] = 1
        self.root_vel = np.zeros((n, 3), F)
        self.root_ang = np.zeros((n, 3), F)
        self.dof_pos = np.zeros((n, num_dof), F)
        self.dof_vel = np.zeros((n, num_dof), F)
        self.target = np.zeros((n, num_dof), F)

    def state(self):
        return (self.root_pos, self.root_rot, self.root_vel, self.root_ang, self.dof_pos, self.dof_vel)

    def set_state(self, env_ids, qpos, qvel):
        self.root_pos[env_ids] = qpos[:, 0:3]
        self.root_rot[env_ids] = qpos[:, 3:7]
        self.dof_pos[env_ids] = qpos[:, 7:]
        self.target[env_ids] = qpos[:, 7:]
        self.root_vel[env_ids] = qvel[:, 0:3]
        self.root_ang[env_ids] = qvel[:, 3:6]
        self.dof_vel[env_ids] = qvel[:, 6:]

    def step(self, action):
        self.target[:] = action
        qn = (self.dof_pos + self.LAG * (self.target - self.dof_pos)).astype(F)
        self.dof_vel = ((qn - self.dof_pos) / F(self.dt)).astype(F)
        self.dof_pos = qn


class AgentCfg:
    # configs/agent/add_g1.yaml
    def __init__(self, **kw):
        self.discount = 0.99
        self.steps_per_iter = 32
        self.update_epochs = 5
        self.batch_size = 4
        self.td_lambda = 0.95
        self.norm_adv_clip = 4.0
        self.disc_reward_scale = 2.0
        self.task_reward_weight = 0.0
        self.disc_reward_weight = 1.0
        self.learning_rate = 1e-4
        self.action_std = 0.05
        self.loss = L.LossCfg()
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


class Draws:
    """Random draws of one iteration, in consumption order.
    noise[t]: [N,29] N(0,1);  resets[t]: dict(ids, segments, jitter) for the envs that are
    done at step t (ascending env id, as nonzero() returns them);  perms: randperm stream."""

    def __init__(self, noise, resets, perms):
        self.noise, self.resets, self.perms = noise, resets, perms


class Agent:
    def __init__(self, cfg, task_cfg, lib, num_envs, params):
        self.cfg, self.n = cfg, num_envs
        kin = lib.kin
        self.task = T.TaskState(task_cfg, lib, num_envs)
        self.sim = KinematicSim(num_envs, kin.num_dof, task_cfg.dt)
        self.model = L.Model(params, cfg.action_std)
        self.opt = L.AdamW(self.model, cfg.learning_rate)
        _, _, a_mean, a_std = kin.action_bounds()
        self.a_norm = L.Normalizer(kin.num_dof, a_mean, a_std)
        self.obs_norm = None
        self.disc_norm = None
        self.ret_buf = np.zeros(num_envs, F)
        self.len_buf = np.zeros(num_envs, np.int64)
        self.episodes, self.mean_return, self.mean_ep_len = 0, F(0), F(0)
        self.obs = None

    def reset_envs(self, env_ids, draw):
        # add_agent.py:221-233
        if len(env_ids) > 0:
            qpos, qvel = self.task.reset(env_ids, draw["ids"], draw["segments"], draw["jitter"])
            self.sim.set_state(env_ids, qpos, qvel)
            self.obs, self.disc_obs, self.disc_demo = self.task.compute_obs(self.sim.state())
        return self.obs

    def init(self, draw):
        self.reset_envs(np.arange(self.n), draw)
        self.obs_norm = L.Normalizer(self.obs.shape[1])
        self.disc_norm = L.DiffNormalizer(self.disc_obs.shape[1])

    def _track_returns(self, r, done):
        # base_agent.py:596-621
        self.ret_buf += r
        self.len_buf += 1
        ids = np.nonzero(done != T.DONE_NULL)[0]
        if len(ids) > 0:
            new_ret = np.mean(self.ret_buf[ids], dtype=F)
            new_len = np.mean(self.len_buf[ids].astype(F), dtype=F)
            cnt = self.episodes + len(ids)
            w_new, w_old = float(len(ids)) / cnt, float(self.episodes) / cnt
            self.mean_return = F(w_new) * new_ret + F(w_old) * self.mean_return
            self.mean_ep_len = F(w_new) * new_len + F(w_old) * self.mean_ep_len
            self.episodes = cnt
            self.ret_buf[ids] = 0
            self.len_buf[ids] = 0

    def rollout(self, draws, contact=None):
        # base_agent.py:379-391
        Tn, n = self.cfg.steps_per_iter, self.n
        buf = {k: [] for k in ("obs", "next_obs", "action", "reward", "done", "a_logp", "rand_action_mask",
                               "disc_obs", "disc_obs_demo", "motion_ids", "motion_times")}
        for t in range(Tn):
            a, logp, _ = L.actor_step(self.model, self.obs_norm, self.a_norm, self.obs, draws.noise[t])
            buf["obs"].append(self.obs.copy())
            buf["action"].append(a)
            buf["a_logp"].append(logp)
            buf["rand_action_mask"].append(np.ones(n, F))
            self.obs_norm.record(self.obs)
            self.sim.step(a)
            c = None if contact is None else contact[t]
            obs, d_obs, d_demo, r, done = self.task.step(self.sim.state(), c)
            self.obs, self.disc_obs, self.disc_demo = obs, d_obs, d_demo
            self._track_returns(r, done)
            buf["next_obs"].append(obs.copy())
            buf["reward"].append(r)
            buf["done"].append(done)
            buf["disc_obs"].append(d_obs)
            buf["disc_obs_demo"].append(d_demo)
            buf["motion_ids"].append(self.task.motion_ids.copy())
            buf["motion_times"].append(self.task.motion_times())
            self.disc_norm.record(d_demo - d_obs)  # add_agent.py:106-108
            ids = np.nonzero(done != T.DONE_NULL)[0]
            self.reset_envs(ids, draws.resets(t, ids) if callable(draws.resets) else draws.resets[t])
        self.buf = {k: np.stack(v) for k, v in buf.items()}
        return self.buf

    def build_train_data(self):
        # add_agent.py:110-133 then ppo_agent.py:111-159
        b, cfg = self.buf, self.cfg
        Tn, n = b["reward"].shape
        flat = lambda x: x.reshape((Tn * n,) + x.shape[2:])
        d_obs, d_demo = flat(b["disc_obs"]), flat(b["disc_obs_demo"])
        with torch.no_grad():
            logits = self.model.disc(L.t32(self.disc_norm.normalize(d_demo - d_obs))).numpy()
        disc_r = L.disc_reward(logits, cfg.disc_reward_scale)
        diff = d_obs - d_demo
        self.task.sampler.update_errors(flat(b["motion_ids"]), flat(b["motion_times"]), np.sum(diff * diff, axis=-1, dtype=F))
        r = (F(cfg.task_reward_weight) * flat(b["reward"]) + F(cfg.disc_reward_weight) * disc_r).astype(F).reshape(Tn, n)
        b["reward"] = r
        with torch.no_grad():
            next_vals = self.model.critic(L.t32(self.obs_norm.normalize(flat(b["next_obs"])))).numpy().reshape(Tn, n)
            vals = self.model.critic(L.t32(self.obs_norm.normalize(flat(b["obs"])))).numpy().reshape(Tn, n)
        next_vals = next_vals.copy()
        next_vals[b["done"] == T.DONE_SUCC] = 0.0  # env.py:181-187 -> both terminal values are 0
        next_vals[b["done"] == T.DONE_FAIL] = 0.0
        ret = L.td_lambda_return(r, next_vals, b["done"], cfg.discount, cfg.td_lambda)
        adv, adv_mean, adv_std = L.advantages(ret, vals, b["rand_action_mask"], cfg.norm_adv_clip)
        b["tar_val"], b["adv"] = ret, adv
        dr64 = disc_r.astype(np.float64)
        return dict(adv_mean=float(adv_mean), adv_std=float(adv_std), disc_reward_mean=float(dr64.mean()),
                    disc_reward_std=float(dr64.std(ddof=1)))

    def update(self, draws):
        # ppo_agent.py:171-192
        b, cfg = self.buf, self.cfg
        Tn, n = b["reward"].shape
        total = Tn * n
        flat = {k: v.reshape((total,) + v.shape[2:]) for k, v in b.items()}
        bs = cfg.batch_size * n
        nb = int(np.ceil(float(total) / bs))
        stream = L.SampleStream(draws.perms)
        acc = {}
        for _ in range(cfg.update_epochs):
            for _ in range(nb):
                idx = stream.sample(bs, total)
                mb = dict(
                    norm_obs=self.obs_norm.normalize(flat["obs"][idx]),
                    norm_action=self.a_norm.normalize(flat["action"][idx]),
                    a_logp=flat["a_logp"][idx], adv=flat["adv"][idx], tar_val=flat["tar_val"][idx],
                    rand_action_mask=flat["rand_action_mask"][idx],
                    norm_diff=self.disc_norm.normalize(flat["disc_obs_demo"][idx] - flat["disc_obs"][idx]),
                )
                loss, info = L.compute_loss(self.model, cfg.loss, mb)
                self.opt.step(loss)
                for k, v in info.items():
                    acc[k] = acc.get(k, 0.0) + v
        steps = cfg.update_epochs * nb
        return {k: v / steps for k, v in acc.items()}

    def train_iter(self, draws, contact=None):
        # base_agent.py:353-374
        self.rollout(draws, contact)
        info = self.build_train_data()
        info.update(self.update(draws))
        self.obs_norm.update()  # amp_agent.py:61-63 via base_agent.py:365-366
        self.disc_norm.update()
        info.update(mean_return=float(self.mean_return), mean_ep_len=float(self.mean_ep_len), num_eps=self.episodes)
        return info

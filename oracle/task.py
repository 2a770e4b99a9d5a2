"""Per-env task layer of the hot path: observation / discriminator-observation assembly,
history ring, imitation reward, done flags, adaptive start-time sampler and masked reset
(oracle restatement of add_gym/learning/add/{add_observation,add_reward,add_done,add_motion}.py,
learning/sampler.py and util/circular_buffer.py).  Test infrastructure, numpy fp32."""
import numpy as np

from . import quat as Q

F = np.float32
DONE_NULL, DONE_FAIL, DONE_SUCC, DONE_TIME = 0, 1, 2, 3  # base_agent.py:16-20


class TaskCfg:
    """The task keys the hot path reads (configs/task/pose.yaml defaults)."""

    def __init__(self, **kw):
        self.dt = 0.01  # configs/engine/genesis.yaml:5
        self.global_obs = True
        self.root_height_obs = True
        self.enable_vel_obs = False
        self.enable_phase_obs = False
        self.num_phase_encoding = 4
        self.enable_tar_obs = True
        self.tar_obs_steps = [1, 2, 3, 4, 5, 6]
        self.num_disc_obs_steps = 3
        self.max_episode_length = 20.0
        self.enable_early_termination = True
        self.pose_termination = True
        self.pose_termination_dist = 1.0
        self.rand_reset = True
        self.num_segments = 20
        self.temperature = None
        self.reward_pose_w, self.reward_vel_w = 0.5, 0.1
        self.reward_root_pose_w, self.reward_root_vel_w = 0.15, 0.1
        self.reward_pose_scale, self.reward_vel_scale = 0.25, 0.01
        self.reward_root_pose_scale, self.reward_root_vel_scale = 5.0, 1.0
        self.dof_err_w = None  # add_reward.py:28-52: per-dof weights of the pose / velocity error sums (None = all ones)
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)

    def track_global_root(self):
        return self.enable_tar_obs and self.global_obs  # add_observation.py:349-350


# ---------------------------------------------------------------- observations
def char_obs(cfg, root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel):
    # add_observation.py:422-459
    heading_inv = Q.calc_heading_quat_inv(root_rot)
    if cfg.global_obs:
        rot_obs = Q.quat_to_tan_norm(root_rot)
    else:
        rot_obs = Q.quat_to_tan_norm(Q.quat_mul(heading_inv, root_rot))
    parts = [rot_obs, dof_pos]
    if cfg.enable_vel_obs:
        if cfg.global_obs:
            parts += [root_vel, root_ang_vel, dof_vel]
        else:
            parts += [Q.quat_rotate(heading_inv, root_vel), Q.quat_rotate(heading_inv, root_ang_vel), dof_vel]
    if cfg.root_height_obs:
        parts = [root_pos[:, 2:3]] + parts
    return np.concatenate(parts, axis=-1).astype(F)


def phase_obs(cfg, phase):
    # add_observation.py:557-575
    out = [phase[:, None]]
    if cfg.num_phase_encoding > 0:
        scale = (F(2.0) * F(np.pi) * np.power(F(2.0), np.arange(cfg.num_phase_encoding, dtype=F))).astype(F)
        val = phase[:, None] * scale[None]
        out += [np.sin(val).astype(F), np.cos(val).astype(F)]
    return np.concatenate(out, axis=-1).astype(F)


def tar_obs(cfg, ref_root_pos, ref_root_rot, tar_pos, tar_rot, tar_dof):
    # add_observation.py:578-650; tar_* are [N,K,.]
    pos_obs = (tar_pos - ref_root_pos[:, None, :]).astype(F)
    rot = tar_rot
    if not cfg.global_obs:
        hinv = np.broadcast_to(Q.calc_heading_quat_inv(ref_root_rot)[:, None, :], tar_rot.shape)
        pos_obs = Q.quat_rotate(hinv, pos_obs)
        rot = Q.quat_mul(hinv, tar_rot)
    if cfg.root_height_obs:
        pos_obs = pos_obs.copy()
        pos_obs[..., 2] = tar_pos[..., 2]
    else:
        pos_obs = pos_obs[..., :2]
    out = np.concatenate([pos_obs, Q.quat_to_tan_norm(rot), tar_dof], axis=-1)
    return out.reshape(out.shape[0], -1).astype(F)


def disc_obs(cfg, root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel):
    # add_observation.py:462-554; inputs are [N,H,.] oldest..newest
    pos_o = root_pos.copy()
    if not cfg.global_obs:
        pos_o[..., 0:2] = 0
    out = [pos_o, Q.quat_to_tan_norm(root_rot), dof_pos]
    if cfg.enable_vel_obs:
        if cfg.global_obs:
            out += [root_vel, root_ang_vel, dof_vel]
        else:
            hinv = Q.calc_heading_quat_inv(root_rot)
            out += [Q.quat_rotate(hinv, root_vel), Q.quat_rotate(hinv, root_ang_vel), dof_vel]
    o = np.concatenate(out, axis=-1)
    return o.reshape(o.shape[0], -1).astype(F)


# ---------------------------------------------------------------- reward / done
def _to_local_root(rot, vel, ang):
    # add_reward.py:91-101
    hinv = Q.calc_heading_quat_inv(rot)
    return Q.quat_mul(hinv, rot), Q.quat_rotate(hinv, vel), Q.quat_rotate(hinv, ang)


def reward(cfg, sim, ref, dof_err_w=None):
    """sim/ref: tuples (root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel).
    add_reward.py:103-177."""
    rp, rr, rv, ra, dp, dv = sim
    tp, tr, tv, ta, tdp, tdv = ref
    w = np.ones(dp.shape[-1], F) if dof_err_w is None else np.asarray(dof_err_w, F)
    pose_err = np.sum(w * (tdp - dp) * (tdp - dp), axis=-1, dtype=F)
    vel_err = np.sum(w * (tdv - dv) * (tdv - dv), axis=-1, dtype=F)
    pd = (tp - rp).astype(F).copy()
    track_root = cfg.track_global_root()
    if not track_root:
        pd[..., 0:2] = 0
    if not cfg.root_height_obs:
        pd[..., 2] = 0
    root_pos_err = np.sum(pd * pd, axis=-1, dtype=F)
    if not track_root:
        rr, rv, ra = _to_local_root(rr, rv, ra)
        tr, tv, ta = _to_local_root(tr, tv, ta)
    rot_err = Q.quat_diff_angle(rr, tr)
    rot_err = rot_err * rot_err
    root_vel_err = np.sum((tv - rv) * (tv - rv), axis=-1, dtype=F)
    root_ang_err = np.sum((ta - ra) * (ta - ra), axis=-1, dtype=F)
    pose_r = np.exp(-F(cfg.reward_pose_scale) * pose_err)
    vel_r = np.exp(-F(cfg.reward_vel_scale) * vel_err)
    root_pose_r = np.exp(-F(cfg.reward_root_pose_scale) * (root_pos_err + F(0.1) * rot_err))
    root_vel_r = np.exp(-F(cfg.reward_root_vel_scale) * (root_vel_err + F(0.1) * root_ang_err))
    r = (F(cfg.reward_pose_w) * pose_r + F(cfg.reward_vel_w) * vel_r
         + F(cfg.reward_root_pose_w) * root_pose_r + F(cfg.reward_root_vel_w) * root_vel_r)
    return r.astype(F)


def done_flags(cfg, time, motion_times, motion_len, loop_mode, root_pos, dof_pos, ref_root_pos, ref_dof_pos, contact):
    # add_done.py:96-147; int32, bit-exact
    done = np.zeros(time.shape[0], np.int32)
    done[time >= F(cfg.max_episode_length)] = DONE_TIME
    done[(motion_times >= motion_len) & (loop_mode != 1)] = DONE_SUCC
    if cfg.enable_early_termination:
        failed = np.zeros(time.shape[0], bool)
        if contact is not None and contact.shape[0] > 0:
            failed |= contact.astype(bool)
        if cfg.pose_termination:
            diff = (ref_dof_pos - dof_pos).astype(F)
            dof_err = np.mean(diff * diff, axis=-1, dtype=F)
            pose_fail = dof_err > F(cfg.pose_termination_dist)
            if cfg.track_global_root():
                rd = (ref_root_pos - root_pos).astype(F)
                pose_fail |= np.sum(rd * rd, axis=-1, dtype=F) > F(cfg.pose_termination_dist)
            failed |= pose_fail
        failed &= time > F(0.0)
        done[failed] = DONE_FAIL
    return done


# ---------------------------------------------------------------- sampler
def torch_floor_divide_f32(a, b):
    """fp32 `a // b` as torch evaluates it (c10::div_floor_floating): fmod-based floor with a
    half-way correction.  sampler.py:88 quantises start times with it."""
    a = np.asarray(a, F)
    b = F(b)
    mod = np.fmod(a, b).astype(F)
    div = ((a - mod) / b).astype(F)
    fix = (mod != 0) & ((b < 0) != (mod < 0))
    div = np.where(fix, div - F(1), div).astype(F)
    fl = np.floor(div).astype(F)
    fl = np.where(div - fl > F(0.5), fl + F(1), fl).astype(F)
    return np.where(div != 0, fl, np.copysign(F(0), a / b)).astype(F)


class SegmentSampler:
    # sampler.py:5-92
    def __init__(self, clip_lengths, dt, num_segments=20, temperature=None, min_start_time=0.0):
        self.num_segments, self.dt, self.temperature = num_segments, dt, temperature
        self.min_start_time = min_start_time
        self.segment_sizes = (np.asarray(clip_lengths, F) / F(num_segments)).astype(F)
        self.errors = np.ones((len(clip_lengths), num_segments), F)

    def update_errors(self, clip_ids, times, track_err):
        # sampler.py:20-55: scatter-mean per (clip, segment), EMA 0.9/0.1 on touched cells
        seg = np.maximum(self.segment_sizes[clip_ids], F(1e-6))
        si = np.clip((np.asarray(times, F) / seg).astype(np.int64), 0, self.num_segments - 1)
        flat = clip_ids * self.num_segments + si
        n = self.errors.size
        cnt = np.bincount(flat, minlength=n).astype(F)
        sm = np.zeros(n, F)
        np.add.at(sm, flat, np.asarray(track_err, F))
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = (sm / cnt).reshape(self.errors.shape)
        upd = (cnt > 0).reshape(self.errors.shape)
        self.errors = np.where(upd, F(0.9) * self.errors + F(0.1) * mean, self.errors).astype(F)

    def probs(self, clip_ids):
        # sampler.py:57-73: softmax(err / (max err over the batch's clips + 1e-6))
        e = self.errors[clip_ids]
        temp = (np.max(e) + F(1e-6)) if self.temperature is None else F(self.temperature)
        z = (e / temp).astype(F)
        z = z - z.max(axis=-1, keepdims=True)
        ez = np.exp(z).astype(F)
        return (ez / ez.sum(axis=-1, keepdims=True, dtype=F)).astype(F)

    def start_time(self, clip_ids, segments, jitter_u):
        # sampler.py:75-92 given the multinomial segment draw and the rand() jitter draw
        seg = self.segment_sizes[clip_ids]
        t = (segments.astype(F) * seg).astype(F)
        t = (t + np.asarray(jitter_u, F) * seg).astype(F)
        t = (torch_floor_divide_f32(t, self.dt) * F(self.dt)).astype(F)
        return np.maximum(t, F(self.min_start_time)).astype(F)


# ---------------------------------------------------------------- env state machine
class TaskState:
    """All per-env state between the engine boundary and the agent, advanced exactly in the
    order of ADDAgent._step_env / _reset_envs (add_agent.py:204-233)."""

    FIELDS = ("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_pos", "dof_vel")

    def __init__(self, cfg, lib, num_envs):
        self.cfg, self.lib, self.n = cfg, lib, num_envs
        self.time = np.zeros(num_envs, F)  # env.py:124 time_buf
        self.motion_ids = np.zeros(num_envs, np.int64)
        self.time_off = np.zeros(num_envs, F)
        self.done = np.zeros(num_envs, np.int32)
        h = cfg.num_disc_obs_steps
        dims = (3, 4, 3, 3, lib.kin.num_dof, lib.kin.num_dof)
        self.hist = {k: np.zeros((num_envs, h, d), F) for k, d in zip(self.FIELDS, dims)}
        self.head = 0  # circular_buffer.py:8 one shared head per ring (all six advance together)
        self.ref = None
        self.sampler = SegmentSampler(lib.lengths, cfg.dt, cfg.num_segments, cfg.temperature,
                                      (cfg.num_disc_obs_steps - 1) * cfg.dt)  # add_motion.py:26-32
        # fp32 time offsets, exactly the tensors the reference builds
        self.tar_dt = (F(cfg.dt) * np.asarray(cfg.tar_obs_steps, F)).astype(F)  # add_observation.py:215
        self.demo_dt = (F(-cfg.dt) * np.arange(h, dtype=F))[::-1].astype(F)  # add_observation.py:366-369

    def motion_times(self):
        return (self.time + self.time_off).astype(F)  # add_observation.py:352-354

    # circular_buffer.py:17-20
    def hist_push(self, sim):
        for k, v in zip(self.FIELDS, sim):
            self.hist[k][:, self.head] = v
        self.head = (self.head + 1) % self.cfg.num_disc_obs_steps

    # circular_buffer.py:46-57
    def hist_all(self):
        return tuple(np.concatenate([self.hist[k][:, self.head:], self.hist[k][:, :self.head]], axis=1) for k in self.FIELDS)

    # circular_buffer.py:22-29
    def hist_fill(self, env_ids, data):
        h = self.cfg.num_disc_obs_steps
        for k, d in zip(self.FIELDS, data):
            self.hist[k][env_ids, :self.head] = d[:, h - self.head:]
            self.hist[k][env_ids, self.head:] = d[:, :h - self.head]

    def update_ref(self):
        self.ref = self.lib.get_step(self.motion_ids, self.motion_times())  # add_observation.py:163-174

    def demo_frames(self, ids, t0):
        # add_observation.py:362-402
        h = self.cfg.num_disc_obs_steps
        t = (t0[:, None] + self.demo_dt[None, :]).astype(F).reshape(-1)
        out = self.lib.get_step(np.repeat(ids, h), t)
        return tuple(o.reshape(ids.shape[0], h, -1) for o in out)

    def compute_obs(self, sim):
        """add_observation.py:231-306 -> (obs, disc_obs, disc_obs_demo)."""
        cfg = self.cfg
        t = self.motion_times()
        parts = [char_obs(cfg, *sim)]
        if cfg.enable_phase_obs:
            parts.append(phase_obs(cfg, self.lib.calc_phase(self.motion_ids, t)))
        if cfg.enable_tar_obs:
            k = len(cfg.tar_obs_steps)
            tt = (t[:, None] + self.tar_dt[None, :]).astype(F).reshape(-1)
            tp, tr, _, _, td, _ = self.lib.get_step(np.repeat(self.motion_ids, k), tt)
            tp, tr, td = (x.reshape(self.n, k, -1) for x in (tp, tr, td))
            if cfg.global_obs:
                rp, rr = sim[0], sim[1]
            else:
                rp, rr = tp[:, 0], tr[:, 0]
            parts.append(tar_obs(cfg, rp, rr, tp, tr, td))
        obs = np.concatenate(parts, axis=-1).astype(F)
        d_obs = disc_obs(cfg, *self.hist_all())
        d_demo = disc_obs(cfg, *self.demo_frames(self.motion_ids, t))
        return obs, d_obs, d_demo

    def step(self, sim, contact, ctrl_dt=None):
        """Everything after scene.step(): time += dt (env.py:155), ref gather, ring push, obs,
        reward, done (add_agent.py:204-219)."""
        self.time = (self.time + F(self.cfg.dt if ctrl_dt is None else ctrl_dt)).astype(F)
        self.update_ref()
        self.hist_push(sim)
        obs, d_obs, d_demo = self.compute_obs(sim)
        r = reward(self.cfg, sim, self.ref, self.cfg.dof_err_w)
        self.done = done_flags(self.cfg, self.time, self.motion_times(), self.lib.lengths[self.motion_ids],
                               self.lib.loop_modes[self.motion_ids], sim[0], sim[4], self.ref[0], self.ref[4], contact)
        return obs, d_obs, d_demo, r, self.done.copy()

    def reset(self, env_ids, new_ids, segments, jitter_u):
        """add_agent.py:221-233 given the random draws.  Returns (qpos[n,36], qvel[n,35]) that
        the engine receives through set_qpos / set_dofs_velocity (add_observation.py:314-331)."""
        self.time[env_ids] = 0  # env.py:161
        self.done[env_ids] = DONE_NULL  # add_done.py:92-93
        if self.cfg.rand_reset:
            times = self.sampler.start_time(new_ids, segments, jitter_u)
        else:
            times = np.zeros(len(env_ids), F)
        self.motion_ids[env_ids] = new_ids
        self.time_off[env_ids] = times
        self.update_ref()
        ref = [x[env_ids] for x in self.ref]
        qpos = np.concatenate([ref[0], ref[1], ref[4]], axis=-1)
        qvel = np.concatenate([ref[2], ref[3], ref[5]], axis=-1)
        t0 = (self.time[env_ids] + self.time_off[env_ids]).astype(F)
        self.hist_fill(env_ids, self.demo_frames(self.motion_ids[env_ids], t0))  # add_observation.py:334-344
        return qpos, qvel

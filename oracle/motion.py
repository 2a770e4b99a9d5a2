"""Reference-motion library: clip ingest, 100 Hz step tables and the O(1) frame lookup
(oracle restatement of add_gym/anim/motion_lib.py and anim/motion.py).  Test infrastructure."""
import numpy as np

from . import quat as Q

F = np.float32
CLIP_FPS = 30  # anim/motion.py:15 default
LOOP_CLAMP, LOOP_WRAP = 0, 1  # anim/motion.py:6-8


def read_motion_csv(path):
    # anim/motion.py:26-31: one frame per line, comma separated, parsed as float64
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([float(v) for v in line.split(",")])
    return np.asarray(rows, np.float64)


def torch_cpu_arange(n, step, vec=8):
    """Values of torch.arange(0, end, step) (fp32, CPU) as the reference's step-table clock
    uses them (motion_lib.py:296-298): ceil(end/step) rows; torch's CPU kernel fills blocks
    of 2*vec lanes as fl32(fl32(step*block_start) + lane*step) (base rounded to fp32 first,
    lane offset added in double) and the tail as fl32(step*i) in double.  vec=8 is what
    torch 2.10 does on this x86 host (checked against torch.arange, tests/test_oracle_*).
    These 1-ulp differences decide which raw frame a few table rows blend from."""
    out = np.empty(n, F)
    nv = (n // (2 * vec)) * (2 * vec)
    lane = np.arange(vec, dtype=np.float64) * step
    for b in range(0, nv, vec):
        out[b:b + vec] = (np.float64(F(step * b)) + lane).astype(F)
    out[nv:] = (np.arange(nv, n, dtype=np.float64) * step).astype(F)
    return out


class MotionLib:
    """frames_list: list of float64 [F,36] arrays (root xyz, root quat xyzw, 29 joint angles in
    `motion_order`); kin: oracle.kin.KinTree; dt: control step."""

    def __init__(self, frames_list, weights, motion_order, kin, dt, reference_compat=True):
        self.kin, self.dt = kin, dt
        self.dt_inv = round(1 / dt)  # motion_lib.py:23
        self.reference_compat = reference_compat
        # motion_lib.py:102-105: column of joint j = index of its name in the clip's order
        reorder = [motion_order.index(n) for n in kin.joint_names[1:]]
        self.num_frames, self.lengths = [], []
        fr = dict(root_pos=[], root_rot=[], root_vel=[], root_ang_vel=[], joint_rot=[], dof_vel=[])
        for frames in frames_list:
            nf = frames.shape[0]
            fps = CLIP_FPS
            # motion_lib.py:10-15, 106-114
            root_pos = frames[:, 0:3].astype(F)
            root_rot = frames[:, [6, 3, 4, 5]].astype(F)
            joint_dof = frames[:, 7:].astype(F)[:, reorder]
            joint_rot = kin.dof_to_rot(joint_dof)
            # motion_lib.py:210-219
            root_vel = np.zeros_like(root_pos)
            root_vel[:-1] = F(fps) * (root_pos[1:] - root_pos[:-1])
            root_vel[-1] = root_vel[-2]
            root_ang = np.zeros_like(root_pos)
            root_ang[:-1] = F(fps) * Q.quat_to_exp_map(Q.quat_diff(root_rot[:-1], root_rot[1:]))
            root_ang[-1] = root_ang[-2]
            dof_vel = kin.frame_dof_vel(joint_rot, 1.0 / fps)  # motion_lib.py:221
            self.num_frames.append(nf)
            self.lengths.append(1.0 / fps * (nf - 1))  # motion_lib.py:202
            for k, v in zip(fr, (root_pos, root_rot, root_vel, root_ang, joint_rot, dof_vel)):
                fr[k].append(v)
        self.frame = {k: np.concatenate(v, axis=0) for k, v in fr.items()}
        self.num_frames = np.asarray(self.num_frames, np.int64)
        self.lengths = np.asarray(self.lengths, F)  # motion_lib.py:252-254 (fp32)
        self.loop_modes = np.zeros(len(frames_list), np.int32)  # CLAMP (motion.py:14)
        w = np.asarray(weights, F)
        self.weights = w / w.sum(dtype=F)  # motion_lib.py:240-243
        # motion_lib.py:280-282: start offsets counted in RAW 30 fps frames
        self.frame_start = np.concatenate([[0], np.cumsum(self.num_frames)[:-1]]).astype(np.int64)
        self._precompute_steps()

    def num_motions(self):
        return len(self.lengths)

    def total_length(self):
        return float(np.sum(self.lengths, dtype=F))

    # motion_lib.py:361-372 (CLAMP branch only: every clip here is CLAMP)
    def calc_phase(self, ids, times):
        return np.clip(np.asarray(times, F) / self.lengths[ids], F(0), F(1)).astype(F)

    # motion_lib.py:61-88, 118-131
    def calc_motion_frame(self, ids, times):
        nf1 = (self.num_frames[ids] - 1).astype(F)
        phase = self.calc_phase(ids, times)
        i0 = (phase * nf1).astype(np.int64)
        i1 = np.minimum(i0 + 1, self.num_frames[ids] - 1)
        blend = (phase * nf1 - i0.astype(F)).astype(F)
        i0 = i0 + self.frame_start[ids]
        i1 = i1 + self.frame_start[ids]
        f = self.frame
        b = blend[:, None]
        root_pos = ((F(1) - b) * f["root_pos"][i0] + b * f["root_pos"][i1]).astype(F)
        root_rot = Q.slerp(f["root_rot"][i0], f["root_rot"][i1], blend)
        joint_rot = Q.slerp(f["joint_rot"][i0], f["joint_rot"][i1], np.broadcast_to(b, b.shape[:1] + (self.kin.num_dof,)))
        dof_pos = self.kin.rot_to_dof(joint_rot)
        # velocities are taken at i0 without blending (motion_lib.py:70-76)
        return root_pos, root_rot, f["root_vel"][i0], f["root_ang_vel"][i0], dof_pos, f["dof_vel"][i0]

    # motion_lib.py:285-320
    def _precompute_steps(self):
        parts = [[] for _ in range(6)]
        self.step_count = []
        for m in range(self.num_motions()):
            n = int(np.ceil(float(self.lengths[m]) / self.dt))
            times = torch_cpu_arange(n, self.dt)
            ids = np.full(n, m, np.int64)
            for p, v in zip(parts, self.calc_motion_frame(ids, times)):
                p.append(v)
            self.step_count.append(n)
        names = ("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_pos", "dof_vel")
        self.step = {k: np.concatenate(p, axis=0).astype(F) for k, p in zip(names, parts)}
        self.step_count = np.asarray(self.step_count, np.int64)
        self.step_start = np.concatenate([[0], np.cumsum(self.step_count)[:-1]]).astype(np.int64)
        self.total_steps = int(self.step_count.sum())

    # motion_lib.py:322-326.  reference_compat=True reproduces the reference exactly: the
    # clip offset added to the 100 Hz row index is the RAW-frame start;
    # the only deviation is a final clamp where the reference would raise an IndexError.
    def step_index(self, ids, times):
        t = np.asarray(times, F)
        frame = (t * F(self.dt_inv)).astype(np.int64)  # fp32 multiply, truncate toward zero
        if self.reference_compat:
            frame = np.clip(frame, 0, self.total_steps - 1)
            idx = frame + self.frame_start[ids]
            return np.minimum(idx, self.total_steps - 1)
        frame = np.clip(frame, 0, self.step_count[ids] - 1)
        return frame + self.step_start[ids]

    # motion_lib.py:322-335
    def get_step(self, ids, times):
        idx = self.step_index(ids, times)
        s = self.step
        return (s["root_pos"][idx], s["root_rot"][idx], s["root_vel"][idx], s["root_ang_vel"][idx],
                s["dof_pos"][idx], s["dof_vel"][idx])

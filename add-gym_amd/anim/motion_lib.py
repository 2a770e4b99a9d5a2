"""Reference-motion library.  One-time host work (torch CPU, same arithmetic as the reference's
anim/motion_lib.py:164-320 so the 100 Hz step tables are bit-identical to its CPU path), then the
tables live in HBM packed as pose[S,36] | vel[S,36] and every per-step lookup is a HIP kernel
(addhip_motion_lookup / fused into addhip_env_step)."""
import os

import numpy as np
import torch
import yaml

from .. import _lib as L
from ..util import quat as Q
from . import motion as motion_io


def torch_cpu_arange(n, step, vec=8):
    """fp32 values of torch.arange(0, end, step) on the CPU (what the reference's table clock is,
    motion_lib.py:296-298), reproduced without depending on the host's vector width: blocks of 2*vec
    lanes are fl32(fl32(step*block) + lane*step), the tail is fl32(step*i) evaluated in double."""
    out = np.empty(n, np.float32)
    nv = (n // (2 * vec)) * (2 * vec)
    lane = np.arange(vec, dtype=np.float64) * step
    for b in range(0, nv, vec):
        out[b:b + vec] = (np.float64(np.float32(step * b)) + lane).astype(np.float32)
    out[nv:] = (np.arange(nv, n, dtype=np.float64) * step).astype(np.float32)
    return torch.from_numpy(out)


class MotionLib:
    CACHE_VERSION = 1
    _HOST_FIELDS = ("host_pose", "host_vel", "_motion_num_frames", "_motion_lengths", "_motion_loop_modes", "_motion_weights", "_step_counts",
                    "_raw_start", "_step_start")

    def __init__(self, motion_file, motion_order, kin_char_model, dt, device, reference_compat=True, frames_list=None, weights=None,
                 cache_dir=None):
        """cache_dir (task.motion_cache_dir): keep the finished 100 Hz step tables on disk, keyed by the clips' bytes, the
        joint order, dt and the kinematic tree, so that later launches skip the ~1.5 s/clip host ingest.  (The reference
        instead rewrites each .motion file as a .pkl next to it, motion_lib.py:164-200: a side effect on the dataset.)"""
        self._device, self._kin, self._dt = device, kin_char_model, dt
        self._dt_inv = round(1 / dt)  # motion_lib.py:23
        self.reference_compat = bool(reference_compat)
        self.from_cache = False
        if frames_list is None and str(motion_file).startswith(("synthetic:", "synthetic_stand:")):
            from .synth import parse_synthetic, synth_clip, synth_stand_clip

            clips, nframes = parse_synthetic(motion_file)
            gen = synth_stand_clip if str(motion_file).startswith("synthetic_stand:") else synth_clip
            frames_list = [gen(kin_char_model, list(motion_order), c, nframes) for c in range(clips)]
            weights = [1.0] * clips
        cache_file = None
        if frames_list is None:
            files, weights = self._fetch_motion_files(motion_file)
            if cache_dir:
                cache_file = os.path.join(cache_dir, self._cache_key(files, weights, list(motion_order)) + ".pt")
                if os.path.exists(cache_file) and self._load_cache(cache_file):
                    self._upload()
                    return
            frames_list = [motion_io.load_motion(f).frames for f in files]
        self._build(frames_list, weights, list(motion_order))
        if cache_file is not None:
            os.makedirs(cache_dir, exist_ok=True)
            tmp = cache_file + f".{os.getpid()}.tmp"
            torch.save({"version": self.CACHE_VERSION, **{k: getattr(self, k) for k in self._HOST_FIELDS}}, tmp)
            os.replace(tmp, cache_file)  # atomic: several ranks may build the same table at once

    def _cache_key(self, files, weights, order):
        import hashlib

        h = hashlib.sha256()
        h.update(repr((self.CACHE_VERSION, float(self._dt), order, [float(w) for w in weights], self._kin.get_joint_order(),
                       [list(map(float, a)) for a in self._kin.joint_axes()])).encode())
        for f in files:
            with open(f, "rb") as fh:
                h.update(hashlib.sha256(fh.read()).digest())
        return h.hexdigest()[:32]

    def _load_cache(self, path):
        try:
            blob = torch.load(path, weights_only=True)  # our own file; tensors only
        except Exception:
            return False
        if blob.get("version") != self.CACHE_VERSION or any(k not in blob for k in self._HOST_FIELDS):
            return False
        for k in self._HOST_FIELDS:
            setattr(self, k, blob[k])
        self.total_steps = int(self._step_counts.sum())
        self.from_cache = True
        return True

    # motion_lib.py:337-358
    @staticmethod
    def _fetch_motion_files(motion_file):
        if os.path.splitext(motion_file)[1] == ".yaml":
            with open(motion_file) as f:
                entries = yaml.safe_load(f)["motions"]
            assert all(e["weight"] >= 0 for e in entries)
            return [e["file"] for e in entries], [e["weight"] for e in entries]
        return [motion_file], [1.0]

    def _ingest_clip(self, frames, col, fps=30):
        """One clip: raw 30 fps frames -> (pose rows, velocity rows, number of frames, length) of its 100 Hz step table
        (motion_lib.py:164-320 for one motion; same torch ops in the same order as the reference, so bit-identical)."""
        kin = self._kin
        root_pos = torch.tensor(frames[:, 0:3], dtype=torch.float32)
        root_rot = torch.tensor(frames[:, [6, 3, 4, 5]], dtype=torch.float32)  # xyzw -> wxyz
        joint_rot = kin.dof_to_rot(torch.tensor(frames[:, 7:], dtype=torch.float32)[:, col])
        root_vel = torch.zeros_like(root_pos)
        root_vel[:-1] = fps * (root_pos[1:] - root_pos[:-1])
        root_vel[-1] = root_vel[-2]
        root_ang = torch.zeros_like(root_pos)
        root_ang[:-1] = fps * Q.exp_map(Q.mul(root_rot[1:], Q.conj(root_rot[:-1])))
        root_ang[-1] = root_ang[-2]
        dof_vel = kin.compute_frame_dof_vel(joint_rot, 1.0 / fps)
        nframes = frames.shape[0]
        length = torch.tensor(1.0 / fps * (nframes - 1), dtype=torch.float32)  # motion_lib.py:202, 252-254
        n = int(np.ceil(float(length) / self._dt))
        t = torch_cpu_arange(n, self._dt)
        phase = torch.clip(t / length, 0.0, 1.0)  # CLAMP clips (motion_lib.py:361-372)
        nf1 = torch.tensor(nframes - 1, dtype=torch.long)
        i0 = (phase * nf1).long()
        i1 = torch.min(i0 + 1, nf1)
        blend = phase * nf1 - i0
        b = blend.unsqueeze(-1)
        pos = (1.0 - b) * root_pos[i0] + b * root_pos[i1]
        rot = Q.slerp(root_rot[i0], root_rot[i1], blend)
        dof = kin.rot_to_dof(Q.slerp(joint_rot[i0], joint_rot[i1], b))
        pose = torch.cat([pos, rot, dof], dim=-1)
        # velocities are not blended (motion_lib.py:70-76)
        vel = torch.cat([root_vel[i0], root_ang[i0], dof_vel[i0], torch.zeros(n, 1)], dim=-1)
        return pose, vel, nframes, float(1.0 / fps * (nframes - 1))

    def _build(self, frames_list, weights, order):
        """Clips are independent: they are ingested by a pool of host threads (torch releases the GIL inside its ops), in clip
        order in the tables.  The arithmetic per clip is untouched, so the tables stay bit-identical to the reference's."""
        import time
        from concurrent.futures import ThreadPoolExecutor

        kin = self._kin
        col = torch.tensor([order.index(n) for n in kin.get_joint_order()[1:]], dtype=torch.long)  # motion_lib.py:102-111
        t0 = time.perf_counter()
        workers = max(1, min(len(frames_list), len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4, 16))
        if workers > 1:
            with ThreadPoolExecutor(max_workers=workers) as pool:
                clips = list(pool.map(lambda f: self._ingest_clip(f, col), frames_list))
        else:
            clips = [self._ingest_clip(f, col) for f in frames_list]
        self.ingest_seconds = time.perf_counter() - t0
        self._motion_num_frames = torch.tensor([c[2] for c in clips], dtype=torch.long)
        self._motion_lengths = torch.tensor([c[3] for c in clips], dtype=torch.float32)
        self._motion_loop_modes = torch.zeros(len(clips), dtype=torch.int32)
        w = torch.tensor(weights, dtype=torch.float32)
        self._motion_weights = w / w.sum()
        raw_start = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(self._motion_num_frames, 0)[:-1]])
        self._step_counts = torch.tensor([c[0].shape[0] for c in clips], dtype=torch.long)
        step_start = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(self._step_counts, 0)[:-1]])
        self.host_pose = torch.cat([c[0] for c in clips], 0).contiguous()
        self.host_vel = torch.cat([c[1] for c in clips], 0).contiguous()
        self._raw_start, self._step_start = raw_start, step_start
        self.total_steps = int(self._step_counts.sum())
        self._upload()

    def _upload(self):
        dev = self._device
        self.pose = self.host_pose.to(dev)
        self.vel = self.host_vel.to(dev)
        start = self._raw_start if self.reference_compat else self._step_start  # motion_lib.py:280-282 quirk
        self._d_start = start.to(torch.int32).to(dev)
        self._d_steps = self._step_counts.to(torch.int32).to(dev)
        self._d_len = self._motion_lengths.to(dev)
        self._d_loop = self._motion_loop_modes.to(dev)
        self.c_struct = L.MotionT(L.ptr(self.pose), L.ptr(self.vel), L.ptr(self._d_start), L.ptr(self._d_steps), L.ptr(self._d_len),
                                  L.ptr(self._d_loop), self.get_num_motions(), self.total_steps, int(self.reference_compat), float(self._dt_inv))

    def get_num_motions(self):
        return self._motion_lengths.shape[0]

    def get_total_length(self):
        return torch.sum(self._motion_lengths).item()

    def get_motion_lengths(self):
        return self._motion_lengths

    def get_motion_weights(self):
        return self._motion_weights

    def get_motion_length(self, motion_ids):
        return self._d_len[motion_ids.long()]

    def get_motion_loop_mode(self, motion_ids):
        return self._d_loop[motion_ids.long()]

    def step_index(self, motion_ids, motion_times):
        """Row indices only (int32, bit-exact vs motion_lib.py:322-326)."""
        n = motion_ids.shape[0]
        idx = torch.empty(n, dtype=torch.int32, device=self._device)
        # keep both converted inputs referenced until the launch is enqueued (a temporary freed between
        # two ptr() calls could be recycled by the caching allocator for the next one)
        ids32, t32 = motion_ids.to(torch.int32).contiguous(), motion_times.to(torch.float32).contiguous()
        L.call("addhip_motion_lookup", self.c_struct, L.ptr(ids32), L.ptr(t32), n, L.ptr(idx), None, None, L.current_stream())
        return idx

    def get_precomputed_motion_step(self, motion_ids, motion_times, packed=False):
        """(root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel) like motion_lib.py:322-335; packed=True returns
        the two [n,36] rows (pose = pos3|quat4|dof29, vel = vel3|ang3|dofvel29|0) instead of the six views."""
        n = motion_ids.shape[0]
        pose = torch.empty(n, L.POSE_W, device=self._device)
        vel = torch.empty(n, L.POSE_W, device=self._device)
        ids32, t32 = motion_ids.to(torch.int32).contiguous(), motion_times.to(torch.float32).contiguous()
        L.call("addhip_motion_lookup", self.c_struct, L.ptr(ids32), L.ptr(t32), n, None, L.ptr(pose), L.ptr(vel), L.current_stream())
        if packed:
            return pose, vel
        return pose[:, 0:3], pose[:, 3:7], vel[:, 0:3], vel[:, 3:6], pose[:, 7:], vel[:, 6:35]

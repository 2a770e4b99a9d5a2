"""Synthetic G1 clips for benchmarks (there are no datasets on the GPU box): F x 36 frames in the `.motion` column
layout (root xyz, root quaternion xyzw, 29 joint angles in task.motion_joint_order).  SURVEY section 8d recipe: root xy
random walk, z = 0.79 + 0.02 sin, yaw-only random-walk orientation, joints = mid + 0.3 * half_range * sin(2 pi f t + phi),
30 fps, seed = 1234 + clip id."""
import numpy as np


def synth_clip(kin, motion_order, clip_id, num_frames=3600, fps=30):
    rng = np.random.RandomState(1234 + clip_id)
    t = np.arange(num_frames) / fps
    xy = np.cumsum(rng.standard_normal((num_frames, 2)) * 0.01, axis=0)
    z = 0.79 + 0.02 * np.sin(2 * np.pi * 0.5 * t)
    yaw = np.cumsum(rng.standard_normal(num_frames) * 0.01)
    quat_xyzw = np.stack([np.zeros_like(yaw), np.zeros_like(yaw), np.sin(yaw / 2), np.cos(yaw / 2)], axis=-1)
    names = kin.get_joint_order()[1:]
    ranges = kin.dof_ranges()
    mid, half = ranges.mean(axis=1), 0.5 * (ranges[:, 1] - ranges[:, 0])
    freq, phase = rng.uniform(0.2, 1.5, len(names)), rng.uniform(0, 2 * np.pi, len(names))
    dof_bfs = mid + 0.3 * half * np.sin(2 * np.pi * freq * t[:, None] + phase)
    col = [names.index(n) for n in motion_order]  # clip columns follow the file's joint order
    return np.concatenate([xy, z[:, None], quat_xyzw, dof_bfs[:, col]], axis=-1)


def parse_synthetic(spec):
    """'synthetic:<clips>x<frames>' -> (clips, frames)"""
    body = spec.split(":", 1)[1]
    c, _, f = body.partition("x")
    return int(c), int(f or 3600)

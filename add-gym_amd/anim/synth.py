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


def synth_stand_clip(kin, motion_order, clip_id, num_frames=900, fps=30, root_height=0.79):
    """A physically feasible clip for the rigid-body engine: the robot stands (legs and waist at the zero pose, root fixed at
    standing height) and swings its arms: shoulder pitch / elbow sinusoids, seed = 4321 + clip id."""
    rng = np.random.RandomState(4321 + clip_id)
    t = np.arange(num_frames) / fps
    names = kin.get_joint_order()[1:]
    dof = np.zeros((num_frames, len(names)))
    for j, n in enumerate(names):
        if "shoulder_pitch" in n or "elbow" in n:
            dof[:, j] = rng.uniform(0.15, 0.35) * np.sin(2 * np.pi * rng.uniform(0.3, 0.8) * t + rng.uniform(0, 2 * np.pi))
        elif "shoulder_roll" in n:
            dof[:, j] = (0.2 if "left" in n else -0.2) + 0.05 * np.sin(2 * np.pi * 0.4 * t)
    root = np.tile([0.0, 0.0, root_height, 0.0, 0.0, 0.0, 1.0], (num_frames, 1))  # xyz, quat xyzw
    col = [names.index(n) for n in motion_order]
    return np.concatenate([root, dof[:, col]], axis=-1)


def parse_synthetic(spec):
    """'synthetic:<clips>x<frames>' or 'synthetic_stand:<clips>x<frames>' -> (clips, frames)"""
    body = spec.split(":", 1)[1]
    c, _, f = body.partition("x")
    return int(c), int(f or 3600)

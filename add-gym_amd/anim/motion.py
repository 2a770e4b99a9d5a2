"""Clip files: `.motion` = one CSV line per 30 fps frame, 36 columns = root xyz, root quaternion
xyzw, 29 joint angles (reference: anim/motion.py:11-54, motion_lib.py:10-15).  Unlike the
reference no `.pkl` is written next to the file (works on read-only asset mounts)."""
import enum

import numpy as np


class LoopMode(enum.Enum):
    CLAMP = 0
    WRAP = 1


class Motion:
    def __init__(self, loop_mode, fps, frames):
        self.loop_mode, self.fps, self.frames = loop_mode, fps, frames

    def get_length(self):
        return float(self.frames.shape[0] - 1) / self.fps


def load_motion(file, loop_mode=LoopMode.CLAMP, fps=30):
    if file.endswith(".npy"):
        return Motion(loop_mode, fps, np.load(file, allow_pickle=False).astype(np.float64))
    if not file.endswith(".motion"):
        raise ValueError(f"unsupported clip file {file!r} (.motion CSV or .npy; pickles are not loaded)")
    with open(file) as f:
        rows = [[float(v) for v in line.strip().split(",")] for line in f if line.strip()]
    return Motion(loop_mode, fps, np.asarray(rows, np.float64))

"""Shared helpers for the tests: the G1 tree / small motion library built from the committed
golden inputs (the reference tree is NOT read: /root/reference does not exist on the GPU box)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G1_XML = os.path.join(ROOT, "add-gym_amd", "assets", "g1_29_kinematics.xml")


def gload(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)


def kin_meta():
    return json.loads(str(gload("kin_tree")["meta"]))


def oracle_kin():
    from oracle.kin import KinTree

    return KinTree(G1_XML)


TABLES = ("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_pos", "dof_vel")


def oracle_lib(two=False, reference_compat=True, golden_tables=False):
    """golden_tables=True swaps in the reference's own step tables so that everything
    downstream of the lookup is compared on bit-identical table rows (the tables themselves
    are pinned in test_motion_tables)."""
    from oracle.motion import MotionLib

    m = gload("motion_small")
    meta = kin_meta()
    if two:
        lib = MotionLib([m["two_frames0"], m["two_frames1"]], [1.0, 3.0], meta["motion_joint_order"], oracle_kin(), 0.01, reference_compat)
    else:
        lib = MotionLib([m["frames"]], [1.0], meta["motion_joint_order"], oracle_kin(), 0.01, reference_compat)
    if golden_tables:
        pre = "two_step_" if two else "step_"
        lib.step = {k: m[pre + k].copy() for k in TABLES}
    return lib


def variant(g, prefix):
    return {k[len(prefix) + 1:]: g[k] for k in g.files if k.startswith(prefix + ".")}


# ---------------------------------------------------------------- GPU-side helpers
DEFAULT_TASK = dict(
    global_obs=True, root_height_obs=True, enable_vel_obs=False, enable_phase_obs=False, enable_tar_obs=True,
    tar_obs_steps=[1, 2, 3, 4, 5, 6], num_disc_obs_steps=3, max_episode_length=20.0, enable_early_termination=True,
    pose_termination=True, pose_termination_dist=1.0, rand_reset=True,
    reward_pose_w=0.5, reward_vel_w=0.1, reward_root_pose_w=0.15, reward_root_vel_w=0.1,
    reward_pose_scale=0.25, reward_vel_scale=0.01, reward_root_pose_scale=5.0, reward_root_vel_scale=1.0,
)


def pack_pose(root_pos, root_rot, dof_pos):
    return np.concatenate([root_pos, root_rot, dof_pos], axis=-1).astype(np.float32)


def pack_vel(root_vel, root_ang, dof_vel):
    z = np.zeros(root_vel.shape[:-1] + (1,), np.float32)
    return np.concatenate([root_vel, root_ang, dof_vel, z], axis=-1).astype(np.float32)


class HipMotion:
    """addhip_motion_t over the golden step tables (device tensors kept alive here)."""

    def __init__(self, two=False, reference_compat=True):
        import torch
        import add_gym_amd._lib as L

        m = gload("motion_small")
        pre = "two_step_" if two else "step_"
        dev = "cuda"
        self.pose = torch.tensor(pack_pose(m[pre + "root_pos"], m[pre + "root_rot"], m[pre + "dof_pos"]), device=dev)
        self.vel = torch.tensor(pack_vel(m[pre + "root_vel"], m[pre + "root_ang_vel"], m[pre + "dof_vel"]), device=dev)
        lengths = m[("two_" if two else "") + "lengths"].astype(np.float32)
        steps = np.asarray([int(np.ceil(float(l) / 0.01)) for l in lengths], np.int32)
        raw_start = m[("two_" if two else "") + "start_idx"].astype(np.int32)
        step_start = np.concatenate([[0], np.cumsum(steps)[:-1]]).astype(np.int32)
        self.start = torch.tensor(raw_start if reference_compat else step_start, device=dev)
        self.steps = torch.tensor(steps, device=dev)
        self.len = torch.tensor(lengths, device=dev)
        self.loop = torch.zeros(len(lengths), dtype=torch.int32, device=dev)
        self.lengths = lengths
        self.c = L.MotionT(L.ptr(self.pose), L.ptr(self.vel), L.ptr(self.start), L.ptr(self.steps), L.ptr(self.len), L.ptr(self.loop),
                           len(lengths), int(self.pose.shape[0]), int(reference_compat), 100.0)


# ---------------------------------------------------------------- plane storage (include/addhip.h: ADDHIP_STORE_BF16X3; csrc/planes.h)
def split3(x):
    """fp32 array -> (hi, mid, lo) fp32 arrays that are bf16 values (low 16 bits zero) with hi + mid + lo == x exactly: the
    truncation split of csrc/planes.h, restated with numpy bit operations."""
    x = np.ascontiguousarray(x, np.float32)
    hi = (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    r1 = (x - hi).astype(np.float32)
    mid = (r1.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    lo = (r1 - mid).astype(np.float32)
    return hi, mid, lo


def to_planes(x, ld=None):
    """[rows, cols] fp32 -> [rows, 3 * ld] uint16 in plane storage (groups of 8 values: hi x 8 | mid x 8 | lo x 8); pad columns 0."""
    x = np.asarray(x, np.float32)
    rows, cols = x.shape
    ld = cols if ld is None else ld
    assert ld % 8 == 0 and ld >= cols
    full = np.zeros((rows, ld), np.float32)
    full[:, :cols] = x
    planes = np.stack([(p.view(np.uint32) >> 16).astype(np.uint16) for p in split3(full)], 0)  # [3, rows, ld]
    return np.ascontiguousarray(planes.reshape(3, rows, ld // 8, 8).transpose(1, 2, 0, 3)).reshape(rows, 3 * ld)


def from_planes(u16, cols=None):
    """[rows, 3 * ld] uint16 plane storage -> ([rows, ld] fp32 sum of the planes, [3, rows, ld] fp32 planes)."""
    u16 = np.asarray(u16, np.uint16)
    rows, w = u16.shape
    ld = w // 3
    planes = (u16.reshape(rows, ld // 8, 3, 8).transpose(2, 0, 1, 3).reshape(3, rows, ld).astype(np.uint32) << 16).view(np.float32)
    val = ((planes[0].astype(np.float64) + planes[1]) + planes[2]).astype(np.float32)  # exact: the three parts do not overlap
    if cols is not None:
        val, planes = val[:, :cols], planes[:, :, :cols]
    return val, planes

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

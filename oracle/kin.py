"""Kinematic tree of the G1 from its MJCF + hinge<->quaternion maps (oracle restatement of
add_gym/anim/kin_char_model.py).  Test infrastructure."""
import xml.etree.ElementTree as ET

import numpy as np

from . import quat as Q

F = np.float32


class KinTree:
    """Bodies in breadth-first order (kin_char_model.py:99-169).  Body 0 is the root
    (free joint, 0 dofs here); body j>=1 owns hinge dof j-1 (kin_char_model.py:379-386)."""

    def __init__(self, xml_path):
        root = ET.parse(xml_path).getroot()
        body0 = root.find("worldbody").find("body")
        self.body_names, self.parents, self.joint_names, axes, ranges = [], [], [], [], []
        queue = [(body0, -1)]
        while queue:
            node, parent = queue.pop(0)
            idx = len(self.body_names)
            self.body_names.append(node.attrib["name"])
            self.parents.append(parent)
            if parent < 0:
                self.joint_names.append("root")
                axes.append([0.0, 0.0, 0.0])
                ranges.append([0.0, 0.0])
            else:
                joints = node.findall("joint")
                assert len(joints) == 1, "only single-hinge bodies are on this path"
                j = joints[0]
                self.joint_names.append(j.attrib["name"])
                axes.append([float(v) for v in j.attrib["axis"].split()])
                ranges.append([float(v) for v in j.attrib["range"].split()])
            for child in node.findall("body"):
                queue.append((child, idx))
        self.axes = np.asarray(axes, F)[1:]  # [29,3] hinge axes, dof order
        self.dof_range = np.asarray(ranges, np.float64)[1:]  # [29,2]
        self.num_dof = self.axes.shape[0]

    # kin_char_model.py:595-639 (hinge branch) followed by quat_pos (motion_lib.py:113-114)
    def dof_to_rot(self, dof):
        dof = np.asarray(dof, F)
        axis = np.broadcast_to(self.axes, dof.shape + (3,))
        return Q.quat_pos(Q.axis_angle_to_quat(axis, dof))

    # kin_char_model.py:56-60, 208-224: twist angle about the hinge axis
    def rot_to_dof(self, rot):
        return Q.quat_twist_angle(rot, np.broadcast_to(self.axes, rot.shape[:-1] + (3,)))

    # kin_char_model.py:226-266
    def frame_dof_vel(self, joint_rot, dt):
        drot = Q.quat_normalize(Q.quat_mul(Q.quat_conjugate(joint_rot[:-1]), joint_rot[1:]))
        em = Q.quat_to_exp_map(drot) / F(dt)
        vel = np.sum(self.axes * em, axis=-1, dtype=F).astype(F)
        return np.concatenate([vel, vel[-1:]], axis=0)

    # robot.py:183-212 (limits of the 29 hinge dofs) and base_agent.py:233-252
    def action_bounds(self):
        lo = self.dof_range[:, 0].astype(F)
        hi = self.dof_range[:, 1].astype(F)
        mid = F(0.5) * (hi + lo)
        scale = np.maximum(np.abs(hi - mid), np.abs(lo - mid)) * F(1.4)
        low, high = mid - scale, mid + scale
        a_mean = F(0.5) * (high + low)
        a_std = F(0.5) * (high - low)
        return low.astype(F), high.astype(F), a_mean.astype(F), a_std.astype(F)

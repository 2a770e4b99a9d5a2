"""Kinematic tree of the robot from its MJCF: bodies breadth-first (= simulator link/dof order,
reference: anim/kin_char_model.py:99-169), one hinge per non-root body.  Host-side, one-time."""
import xml.etree.ElementTree as ET

import numpy as np
import torch

from ..util import quat as Q


class KinCharModel:
    def __init__(self, device="cpu"):
        self._device = device

    def load_char_file(self, char_file):
        body0 = ET.parse(char_file).getroot().find("worldbody").find("body")
        self._body_names, self._parents, self._joint_names, axes, ranges = [], [], [], [], []
        todo = [(body0, -1)]
        while todo:
            node, parent = todo.pop(0)
            me = len(self._body_names)
            self._body_names.append(node.attrib["name"])
            self._parents.append(parent)
            hinges = [j for j in node.findall("joint") if j.attrib.get("type", "hinge") == "hinge"]
            if parent < 0:
                self._joint_names.append("root")
            else:
                if len(hinges) != 1:
                    raise ValueError(f"body {node.attrib['name']}: exactly one hinge joint per body is supported")
                j = hinges[0]
                self._joint_names.append(j.attrib["name"])
                axes.append([float(v) for v in j.attrib["axis"].split()])
                ranges.append([float(v) for v in j.attrib["range"].split()])
            todo += [(c, me) for c in node.findall("body")]
        self._axes = torch.tensor(axes, dtype=torch.float32)  # [D,3] in dof order
        self._ranges = np.asarray(ranges, np.float64)         # [D,2]

    # surface used by the task layer (same names as the reference class)
    def get_body_names(self):
        return self._body_names

    def get_joint_order(self):
        return self._joint_names

    def get_num_joints(self):
        return len(self._joint_names)

    def get_dof_size(self):
        return self._axes.shape[0]

    def get_body_id(self, name):
        return self._body_names.index(name)

    def joint_axes(self):
        """[D,3] hinge axes in dof order (part of the identity of a step table built with this tree)."""
        return self._axes.tolist()

    def dof_ranges(self):
        return self._ranges

    # hinge angle -> joint quaternion (kin_char_model.py:595-639) made w-positive (motion_lib.py:113-114)
    def dof_to_rot(self, dof):
        axis = torch.broadcast_to(self._axes, dof.shape + (3,))
        return Q.positive(Q.from_axis_angle(axis, dof))

    # joint quaternion -> hinge angle (kin_char_model.py:56-60, 208-224)
    def rot_to_dof(self, rot):
        return Q.twist_angle(rot, self._axes)

    # per-frame finite-difference joint velocities, last frame repeats (kin_char_model.py:226-266)
    def compute_frame_dof_vel(self, joint_rot, dt):
        d = Q.normalized(Q.mul(Q.conj(joint_rot[:-1]), joint_rot[1:]))
        vel = torch.sum(self._axes * (Q.exp_map(d) / dt), dim=-1)
        return torch.cat([vel, vel[-1:]], dim=0)

    # action bounds: 1.4 x the larger half-range about the joint mid-point (robot.py:183-212)
    def action_bounds(self):
        lo = torch.tensor(self._ranges[:, 0], dtype=torch.float32)
        hi = torch.tensor(self._ranges[:, 1], dtype=torch.float32)
        mid = 0.5 * (hi + lo)
        scale = torch.maximum(torch.abs(hi - mid), torch.abs(lo - mid)) * 1.4
        return mid - scale, mid + scale

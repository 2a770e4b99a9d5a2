"""Robot wrapper between the agent and the engine plugin API: entity creation, joint limits, PD gains by joint
family, action bounds, state accessors, ground-contact predicate.  Same attribute/method names as the reference's
add_gym/robot.py (Manipulator) for the members the imitation path uses."""
import re

import torch

from .anim import kin_char_model


class Manipulator:
    GAINS = (("ankle", 120.0), ("knee", 120.0), ("hip", 80.0), ("waist", 50.0), ("shoulder", 50.0), ("elbow", 50.0), ("wrist", 50.0), ("hand", 20.0))

    def __init__(self, num_envs, scene, engine, robot_cfg, env_spacing, enable_ref=False, device="cpu"):
        self._device, self._scene, self._engine, self._num_envs, self._args = device, scene, engine, num_envs, robot_cfg
        path = robot_cfg["urdf_path"]
        self._kin_char_model = kin_char_model.KinCharModel(device)
        self._kin_char_model.load_char_file(path)
        self._robot_entity = scene.add_entity(morph_type="urdf" if path.endswith(".urdf") else "mjcf", morph_file=path, morph_pos=(0.0, 0.0, 0.0),
                                              morph_quat=(1.0, 0.0, 0.0, 0.0), material_type="rigid")
        self._ref_entity = None
        limits = []
        for j in self._robot_entity.joints:
            limits.extend(j.dofs_limit)
        self.joint_limits = torch.tensor(limits, dtype=engine.tc_float)
        self.gain_scale = robot_cfg.get("gain_scale", 1.0)
        base = [d for j in self._robot_entity.joints if re.fullmatch("root_joint|floating_base_joint", j.name) for d in j.dofs_idx]
        self.non_root_joints = [i for i in range(self._robot_entity.n_dofs) if i not in base]

    def on_build(self):
        # PD gains by joint family x gain_scale, kv = 2 sqrt(kp) (robot.py:133-163)
        kp = torch.full((self._robot_entity.n_dofs,), 100.0)
        for j in self._robot_entity.joints:
            for key, val in self.GAINS:
                if key in j.name:
                    kp[list(j.dofs_idx)] = val
                    break
        kp = kp * self.gain_scale
        self._robot_entity.set_dofs_kp(kp)
        self._robot_entity.set_dofs_kv(2.0 * torch.sqrt(kp))

    def get_action_space(self):
        """[29,2] (low, high) = joint mid-point -/+ 1.4 x larger half-range (robot.py:183-212)."""
        lo, hi = self._kin_char_model.action_bounds()
        return torch.stack([lo, hi], dim=1)

    def get_ground_contact_forces_v2(self, surface_plane, contact_idx):
        """True where any valid ground contact involves a link listed in contact_idx (robot.py:221-231)."""
        c = self._robot_entity.get_contacts(with_entity=surface_plane, exclude_self_contact=True)
        a = torch.isin(c["link_a"], contact_idx) & c["valid_mask"]
        b = torch.isin(c["link_b"], contact_idx) & c["valid_mask"]
        return a.any(dim=1) | b.any(dim=1)

    def apply_action(self, action, allowed_action_idx=None):
        if not hasattr(self._robot_entity, "hot_state"):
            action = action[:, :len(self.non_root_joints)]
        self._robot_entity.control_dofs_position(position=action, dofs_idx_local=allowed_action_idx if allowed_action_idx else self.non_root_joints)

    base_pos = property(lambda s: s._robot_entity.get_pos())
    base_quat = property(lambda s: s._robot_entity.get_quat())
    base_lin_vel = property(lambda s: s._robot_entity.get_vel())
    base_ang_vel = property(lambda s: s._robot_entity.get_ang())
    dof_pos = property(lambda s: s._robot_entity.get_dofs_position()[:, 6:])
    dof_vel = property(lambda s: s._robot_entity.get_dofs_velocity()[:, 6:])
    entity = property(lambda s: s._robot_entity)
    ref_entity = property(lambda s: s._ref_entity)

"""KinematicEngine: a synthetic simulator behind the engine plugin API, resident on the GPU.

NOT a reference component and not physics: the reference delegates rigid-body dynamics to Genesis
(out of scope).  Joints move `lag` of the way to their PD target per
control step, the root keeps the state written at reset, there are no contacts (tests may force a
contact flag).  Its purpose is to feed the hot path with simulator state of the right shape so that
rollout+update throughput can be measured (BASELINE.md section 2 uses the same stand-in on the CPU).

State layout is the hot path's own (pose[N,36] | vel[N,36], include/addhip.h) so the agent reads
it in place through `hot_state()`; the BaseEntity getters expose the reference's shapes for any
other caller.
"""
from typing import Dict, List, Optional, Tuple

import torch

from .. import _lib as L
from ..anim.kin_char_model import KinCharModel
from .base_engine import BaseCamera, BaseEngine, BaseEntity, BaseJoint, BaseLink, BaseScene


class _Link(BaseLink):
    def __init__(self, name, idx, idx_local):
        self._v = (name, idx, idx_local)
    name = property(lambda s: s._v[0])
    idx = property(lambda s: s._v[1])
    idx_local = property(lambda s: s._v[2])


class _Joint(BaseJoint):
    def __init__(self, name, dofs, limits):
        self._v = (name, list(dofs), list(limits))
    name = property(lambda s: s._v[0])
    dofs_idx = property(lambda s: s._v[1])
    dofs_idx_local = property(lambda s: s._v[1])
    dofs_limit = property(lambda s: s._v[2])


class _Camera(BaseCamera):
    _in_recording = False
    def follow_entity(self, entity): pass
    def start_recording(self): pass
    def stop_recording(self, filename, fps=30): pass
    def render(self): pass


class KinematicPlane:
    def __init__(self, link_base):
        self._links = [_Link("plane", link_base, 0)]
    links = property(lambda s: s._links)


class KinematicEntity(BaseEntity):
    def __init__(self, scene, morph_file, link_base, lag):
        kin = KinCharModel()
        kin.load_char_file(morph_file)
        self._scene, self._lag = scene, lag
        self._links = [_Link(n, link_base + i, i) for i, n in enumerate(kin.get_body_names())]
        inf = float("inf")
        self._joints = [_Joint("floating_base_joint", range(6), [(-inf, inf)] * 6)]
        for d, (name, rng) in enumerate(zip(kin.get_joint_order()[1:], kin.dof_ranges())):
            self._joints.append(_Joint(name, [6 + d], [(float(rng[0]), float(rng[1]))]))
        self._n_dofs = 6 + kin.get_dof_size()
        self._target = None

    def build(self, n, device):
        self.n, self._device = n, device
        self.pose = torch.zeros(n, L.POSE_W, device=device)
        self.pose[:, 2] = 0.793
        self.pose[:, 3] = 1.0
        self.vel = torch.zeros(n, L.POSE_W, device=device)
        self.forced_contact = torch.zeros(n, dtype=torch.uint8, device=device)  # test hook: non-foot ground contact flag
        self._own_target = torch.zeros(n, 32, device=device)
        self._target = self._own_target

    def hot_state(self):
        """(pose[N,36], vel[N,36], contact[N] u8) read/written in place by the HIP hot path."""
        return self.pose, self.vel, self.forced_contact

    def _rows(self, envs_idx):
        return slice(None) if envs_idx is None else envs_idx

    def get_pos(self): return self.pose[:, 0:3]
    def get_quat(self): return self.pose[:, 3:7]
    def get_vel(self): return self.vel[:, 0:3]
    def get_ang(self): return self.vel[:, 3:6]
    def set_pos(self, pos, envs_idx=None): self.pose[self._rows(envs_idx), 0:3] = pos
    def set_quat(self, quat, envs_idx=None): self.pose[self._rows(envs_idx), 3:7] = quat

    def get_dofs_position(self):
        return torch.cat([self.pose[:, 0:3], torch.zeros(self.n, 3, device=self._device), self.pose[:, 7:]], dim=-1)

    def get_dofs_velocity(self):
        return self.vel[:, 0:35].clone()

    def set_dofs_position(self, position, envs_idx=None, dofs_idx_local=None):
        rows = self._rows(envs_idx)
        cols = list(range(self._n_dofs)) if dofs_idx_local is None else list(dofs_idx_local)
        joint_cols = [c for c in cols if c >= 6]
        if joint_cols:
            sel = [i for i, c in enumerate(cols) if c >= 6]
            tmp = self.pose[rows]
            tmp[:, [1 + c for c in joint_cols]] = position[:, sel]
            self.pose[rows] = tmp

    def set_dofs_velocity(self, velocity, envs_idx=None):
        self.vel[self._rows(envs_idx), 0:35] = velocity

    def set_qpos(self, qpos, envs_idx=None):
        self.pose[self._rows(envs_idx)] = qpos

    def control_dofs_position(self, position, dofs_idx_local=None):
        if position.shape[-1] == 32 and position.is_contiguous():
            self._target = position  # zero-copy: the agent's action slot [N,32]
        else:
            self._own_target[:, :position.shape[-1]] = position
            self._target = self._own_target

    def set_dofs_kp(self, kp): self._kp = kp
    def set_dofs_kv(self, kv): self._kv = kv
    def zero_all_dofs_velocity(self, envs_idx=None): self.vel[self._rows(envs_idx)] = 0

    def get_links_pos(self): return self.pose[:, None, 0:3].expand(-1, len(self._links), -1)
    def get_links_quat(self): return self.pose[:, None, 3:7].expand(-1, len(self._links), -1)
    def get_links_net_contact_force(self): return torch.zeros(self.n, len(self._links), 3, device=self._device)

    def get_contacts(self, with_entity=None, exclude_self_contact=False) -> Dict[str, torch.Tensor]:
        valid = (self.forced_contact != 0)[:, None]
        torso = next(l.idx for l in self._links if l.name == "torso_link")
        link_a = torch.full((self.n, 1), torso, dtype=torch.long, device=self._device)
        return {"link_a": link_a, "link_b": torch.zeros_like(link_a), "valid_mask": valid}

    def get_AABB(self):
        box = torch.zeros(self.n, 2, 3, device=self._device)
        box[:, 0] = torch.tensor([-0.2, -0.3, 0.0], device=self._device)
        box[:, 1] = torch.tensor([0.2, 0.3, 1.3], device=self._device)
        return box

    def get_joint(self, name): return next(j for j in self._joints if j.name == name)
    def get_link(self, name): return next(l for l in self._links if l.name == name)
    joints = property(lambda s: s._joints)
    links = property(lambda s: s._links)
    n_dofs = property(lambda s: s._n_dofs)

    def step(self, dt):
        L.call("addhip_kin_engine_step", L.ptr(self.pose), L.ptr(self.vel), L.ptr(self._target), int(self._target.shape[-1]), self.n,
               float(self._lag), float(dt), L.current_stream())


class KinematicScene(BaseScene):
    def __init__(self, dt, device, lag):
        self._dt, self._device, self._lag, self._t = dt, device, lag, 0
        self._entities: List[KinematicEntity] = []
        self._n_links = 0

    def add_entity(self, morph_type, morph_file=None, morph_pos=(0.0, 0.0, 0.0), morph_quat=(1.0, 0.0, 0.0, 0.0),
                   material_type="rigid", visualize_contact=False):
        if morph_type == "plane":
            e = KinematicPlane(self._n_links)
            self._n_links += 1
            return e
        e = KinematicEntity(self, morph_file, self._n_links, self._lag)
        self._n_links += len(e.links)
        self._entities.append(e)
        return e

    def add_camera(self, res=(640, 480), pos=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, 0.0), fov=40):
        return _Camera()

    def build(self, n_envs, env_spacing: Tuple[float, float] = (1.0, 1.0)):
        for e in self._entities:
            e.build(n_envs, self._device)

    def step(self):
        for e in self._entities:
            e.step(self._dt)
        self._t += 1

    t = property(lambda s: s._t)


class KinematicEngine(BaseEngine):
    def __init__(self, lag: float = 0.5, **cfg):
        self._lag, self.cfg = lag, cfg
        self._device = None

    def init(self, backend: str, precision: str) -> None:
        if backend != "gpu" or not torch.cuda.is_available():
            raise RuntimeError("KinematicEngine runs on the GPU only (its step is a HIP kernel); no CPU fallback")
        self._device = torch.device("cuda", torch.cuda.current_device())
        L.load()

    def create_scene(self, show_viewer, sim_options, rigid_options, vis_options=None, viewer_options=None):
        return KinematicScene(sim_options["dt"], self._device, self._lag)

    device = property(lambda s: s._device)
    tc_float = property(lambda s: torch.float32)

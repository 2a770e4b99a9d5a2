"""RigidBodyEngine: articulated rigid-body dynamics of the robot with PD-controlled joints and ground contact, resident on
the GPU, behind the engine plugin API (engine/base_engine.py == add-gym's add_gym/engine/base_engine.py:93-510).

This is the simulator seat the reference fills with Genesis (engine/genesis_engine.py) or MuJoCo-Warp
(engine/mjwarp_engine.py).  The dynamics are this repo's own (csrc/rigid.hip, restated in float64 in oracle/rigid.py):
articulated-body algorithm with a floating base, `substeps` physics steps per control step, stable-PD joints with
the robot's gains (robot.py:133-163), MJCF joint damping / armature / torque limits, joint-limit springs, spring-damper
ground contacts with regularised Coulomb friction on per-link collision spheres -- all stiff terms integrated implicitly
inside the one O(bodies) sweep.  Semantics kept from the reference's engines: position targets are clamped to the joint range
minus `position_limit_margin`, torques to `max_torque` (mjwarp_engine.py:807-851, 1554-1611); get_contacts reports the links
touching the ground as {link_a, link_b, valid_mask} (mjwarp_engine.py:896-986).  Not modelled: self-collision, joint friction
loss, terrain other than the z = 0 plane.

State lives in the hot path's packed rows (pose[N,36] | vel[N,36], include/addhip.h) and is shared in place through
hot_state(); the BaseEntity getters expose the reference's shapes for any other caller."""
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib as L
from .base_engine import BaseCamera, BaseEngine, BaseEntity, BaseScene
from .kinematic_engine import KinematicPlane, _Camera, _Joint, _Link
from .rigid_model import RigidModelTables


class RigidEntity(BaseEntity):
    def __init__(self, scene, morph_file, link_base, opts):
        self._scene, self._opts = scene, opts
        self.tables = RigidModelTables(morph_file)
        t = self.tables
        self._link_base = link_base
        self._links = [_Link(n, link_base + i, i) for i, n in enumerate(t.names)]
        inf = float("inf")
        self._joints = [_Joint("floating_base_joint", range(6), [(-inf, inf)] * 6)]
        import xml.etree.ElementTree as ET

        jn = {}
        for b in ET.parse(morph_file).getroot().iter("body"):
            for j in b.findall("joint"):
                if j.attrib.get("type", "hinge") == "hinge":
                    jn[b.attrib["name"]] = j.attrib["name"]
        for k_bfs in range(1, t.num_bodies):
            k = t.bfs_of_traversal.index(k_bfs)
            lo, hi = float(t.body[k, 22]), float(t.body[k, 23])
            self._joints.append(_Joint(jn[t.names[k_bfs]], [6 + k_bfs - 1], [(lo, hi)]))
        self._n_dofs = 6 + t.num_bodies - 1
        self._target = None
        self._term_mask = 0
        self._dirty = True

    # ---- build / model upload
    def build(self, n, device):
        self.n, self._device = n, device
        self.pose = torch.zeros(n, L.POSE_W, device=device)
        self.pose[:, 2] = 0.793
        self.pose[:, 3] = 1.0
        self.vel = torch.zeros(n, L.POSE_W, device=device)
        self.contact = torch.zeros(n, dtype=torch.uint8, device=device)      # termination flag of the last step
        self.contact_bits = torch.zeros(n, dtype=torch.int32, device=device)  # one bit per link
        self._own_target = torch.zeros(n, 32, device=device)
        self._target = self._own_target
        # domain randomisation (build-defined extension, off by default; the reference has none): per-env PD gain scale and
        # ground friction, redrawn every `resample_interval` control steps, plus random horizontal pushes of the root.  Drawn on the
        # device by addhip_rigid_randomize from a device-resident control-step counter: part of a captured rollout like any other launch
        dr = self._opts.get("domain_randomization") or {}
        self._dr = dict(dr) if dr.get("enabled", False) else None
        self.env_scale = None
        self._steps = 0
        if self._dr is not None:
            d = self._dr
            self.env_scale = torch.ones(n, 2, device=device)
            self._d_steps = torch.zeros(2, dtype=torch.int64, device=device)  # [control-step index, ticket scratch]
            g0, g1 = d.get("gain_scale", [1.0, 1.0])
            f0, f1 = d.get("friction", [self._opts["friction"]] * 2)
            self._dr_c = L.RigidDrT(int(d.get("seed", 0)), int(d.get("resample_interval", 0)), int(d.get("push_interval", 0)), float(g0), float(g1),
                                    float(f0), float(f1), float(d.get("push_velocity", 0.5)))
            L.call("addhip_rigid_randomize", self._dr_c, L.ptr(self.env_scale), L.ptr(self.vel), n, L.ptr(self._d_steps), 0, L.current_stream())
        self._tables_on_device = None
        self._upload()

    def _upload(self):
        """Model tables -> device.  The device buffers are allocated ONCE and refreshed in place, so the addresses a captured rollout
        has baked into its launches stay valid when gains change; the scalar fields of c_struct (termination mask, step sizes) are
        launch arguments by value, so `model_version` tells a holder of captured launches to re-capture."""
        t, o, dev = self.tables, self._opts, self._device
        # four lanes per env (one per chain of the tree) unless the tree does not fit or the option asks for the one-lane kernel
        chains = t.chain_table() if int(o.get("lanes_per_env", 4)) == 4 else None
        host = dict(body=t.body, topo=t.topo, points=t.points if t.num_points else np.zeros((1, 4), np.float32), chains=chains)
        if self._tables_on_device is None:
            self._tables_on_device = {k: (torch.tensor(v, device=dev) if v is not None else None) for k, v in host.items()}
        else:
            for k, v in host.items():
                d = self._tables_on_device[k]
                if (v is None) != (d is None) or (v is not None and tuple(v.shape) != tuple(d.shape)):
                    raise RuntimeError(f"rigid model table '{k}' changed shape after build")
                if v is not None:
                    d.copy_(torch.as_tensor(v))
        dv = self._tables_on_device
        self._d_body, self._d_topo, self._d_points, self._d_chains = dv["body"], dv["topo"], dv["points"], dv["chains"]
        self.c_struct = L.RigidModelT(t.num_bodies, t.num_points, L.ptr(self._d_body), L.ptr(self._d_topo), L.ptr(self._d_points),
                                      float(o["dt"]), int(o["substeps"]), float(o["gravity"]), float(o["contact_stiffness"]),
                                      float(o["contact_damping"]), float(o["friction"]), float(o["friction_vel_eps"]),
                                      float(o["limit_stiffness"]), float(o["max_torque"]), float(o["position_limit_margin"]), int(self._term_mask),
                                      L.ptr(self.env_scale), L.ptr(self._d_chains))
        self.model_version = getattr(self, "model_version", 0) + 1
        self._dirty = False

    def set_termination_links(self, allowed_link_names):
        """Links whose ground contact is allowed (task.contact_bodies, add_done.py:36-45); contact of any OTHER link raises the
        per-env flag of hot_state() -- the predicate of Manipulator.get_ground_contact_forces_v2 (robot.py:221-231), evaluated
        in the step kernel."""
        allowed = {self.get_link(n).idx_local for n in allowed_link_names}
        self._term_mask = self.tables.link_mask([i for i in range(self.tables.num_bodies) if i not in allowed])
        self._dirty = True

    def hot_state(self):
        """(pose[N,36], vel[N,36], contact[N] u8) read/written in place by the HIP hot path."""
        return self.pose, self.vel, self.contact

    def _rows(self, envs_idx):
        return slice(None) if envs_idx is None else envs_idx

    # ---- BaseEntity state access (reference shapes)
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

    def set_dofs_kp(self, kp):
        self._kp = torch.as_tensor(kp, dtype=torch.float32).cpu()
        self._set_gains()

    def set_dofs_kv(self, kv):
        self._kv = torch.as_tensor(kv, dtype=torch.float32).cpu()
        self._set_gains()

    def _set_gains(self):
        if hasattr(self, "_kp") and hasattr(self, "_kv"):
            self.tables.set_gains(self._kp[6:].numpy(), self._kv[6:].numpy())  # dofs 0-5 = floating base: never actuated
            self._dirty = True

    def zero_all_dofs_velocity(self, envs_idx=None): self.vel[self._rows(envs_idx)] = 0

    # ---- links (forward kinematics on demand; outside the hot path)
    def _fk(self):
        t = self.tables
        q = self.pose[:, 7:]
        w, x, y, z = self.pose[:, 3], self.pose[:, 4], self.pose[:, 5], self.pose[:, 6]
        R0 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                          2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).view(-1, 3, 3)
        R, p = {0: R0}, {0: self.pose[:, 0:3]}
        for k in range(1, t.num_bodies):
            par, ax, dof, link = int(t.topo[k, 0]), int(t.topo[k, 1]), int(t.topo[k, 2]), int(t.topo[k, 7])
            plink = int(t.topo[par, 7])
            Rf = torch.tensor(t.body[k, 3:12].reshape(3, 3), device=self._device)
            r = torch.tensor(t.body[k, 0:3], device=self._device)
            c, s = torch.cos(q[:, dof]), torch.sin(q[:, dof])
            o, zz = torch.ones_like(c), torch.zeros_like(c)
            rows = {0: [o, zz, zz, zz, c, -s, zz, s, c], 1: [c, zz, s, zz, o, zz, -s, zz, c], 2: [c, -s, zz, s, c, zz, zz, zz, o]}[ax]
            Rj = torch.stack(rows, -1).view(-1, 3, 3)
            R[link] = R[plink] @ Rf @ Rj
            p[link] = p[plink] + (R[plink] @ r)
        n = t.num_bodies
        return torch.stack([R[i] for i in range(n)], 1), torch.stack([p[i] for i in range(n)], 1)

    def get_links_pos(self): return self._fk()[1]

    def get_links_quat(self):
        R = self._fk()[0]
        w = torch.sqrt(torch.clamp(1 + R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2], min=1e-12)) / 2
        return torch.stack([w, (R[..., 2, 1] - R[..., 1, 2]) / (4 * w), (R[..., 0, 2] - R[..., 2, 0]) / (4 * w), (R[..., 1, 0] - R[..., 0, 1]) / (4 * w)], -1)

    def get_links_net_contact_force(self): return torch.zeros(self.n, len(self._links), 3, device=self._device)

    def get_contacts(self, with_entity=None, exclude_self_contact=False) -> Dict[str, torch.Tensor]:
        """One slot per link: valid where that link touched the ground plane in the last step (mjwarp_engine.py:896-986 shape)."""
        nl = len(self._links)
        ids = torch.arange(nl, device=self._device)
        valid = ((self.contact_bits[:, None] >> ids[None, :]) & 1).bool()
        link_a = (ids + self._link_base)[None, :].expand(self.n, nl).contiguous()
        plane = with_entity.links[0].idx if with_entity is not None and hasattr(with_entity, "links") else 0
        return {"link_a": link_a, "link_b": torch.full_like(link_a, plane), "valid_mask": valid}

    def get_AABB(self):
        R, p = self._fk()
        t = self.tables
        lo = torch.full((self.n, 3), float("inf"), device=self._device)
        hi = -lo
        pts = torch.tensor(t.points, device=self._device)
        for k in range(t.num_bodies):
            s, c, link = int(t.topo[k, 5]), int(t.topo[k, 6]), int(t.topo[k, 7])
            if c == 0:
                continue
            w = p[:, link, None, :] + torch.einsum("nij,pj->npi", R[:, link], pts[s:s + c, :3])
            rad = pts[s:s + c, 3][None, :, None]
            lo = torch.minimum(lo, (w - rad).amin(1))
            hi = torch.maximum(hi, (w + rad).amax(1))
        return torch.stack([lo, hi], 1)

    def get_joint(self, name): return next(j for j in self._joints if j.name == name)
    def get_link(self, name): return next(l for l in self._links if l.name == name)
    joints = property(lambda s: s._joints)
    links = property(lambda s: s._links)
    n_dofs = property(lambda s: s._n_dofs)

    def step(self):
        if self._dirty:
            self._upload()
        if self._dr is not None:  # redraws / pushes due at this control step, then the device counter advances
            L.call("addhip_rigid_randomize", self._dr_c, L.ptr(self.env_scale), L.ptr(self.vel), self.n, L.ptr(self._d_steps), 1, L.current_stream())
        self._steps += 1
        L.call("addhip_rigid_step", self.c_struct, L.ptr(self.pose), L.ptr(self.vel), L.ptr(self._target), int(self._target.shape[-1]), self.n,
               L.ptr(self.contact), L.ptr(self.contact_bits), L.current_stream())


class RigidScene(BaseScene):
    def __init__(self, device, opts):
        self._device, self._opts, self._t = device, opts, 0
        self._entities: List[RigidEntity] = []
        self._n_links = 0

    def add_entity(self, morph_type, morph_file=None, morph_pos=(0.0, 0.0, 0.0), morph_quat=(1.0, 0.0, 0.0, 0.0),
                   material_type="rigid", visualize_contact=False):
        if morph_type == "plane":
            e = KinematicPlane(self._n_links)
            self._n_links += 1
            return e
        e = RigidEntity(self, morph_file, self._n_links, self._opts)
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
            e.step()
        self._t += 1

    t = property(lambda s: s._t)


class RigidBodyEngine(BaseEngine):
    DEFAULTS = dict(substeps=4, gravity=9.81, contact_stiffness=2.0e4, contact_damping=3.0e2, friction=1.0, friction_vel_eps=0.01,
                    limit_stiffness=2.0e3, max_torque=200.0, position_limit_margin=1e-4, domain_randomization=None, lanes_per_env=4)

    def __init__(self, **cfg):
        self.cfg = cfg
        self._opts = {k: cfg.get(k, v) for k, v in self.DEFAULTS.items()}
        self._device = None

    def init(self, backend: str, precision: str) -> None:
        if backend != "gpu" or not torch.cuda.is_available():
            raise RuntimeError("RigidBodyEngine runs on the GPU only (its step is a HIP kernel); no CPU fallback")
        self._device = torch.device("cuda", torch.cuda.current_device())
        L.load()

    def create_scene(self, show_viewer, sim_options, rigid_options, vis_options=None, viewer_options=None):
        opts = dict(self._opts, dt=float(sim_options["dt"]))
        return RigidScene(self._device, opts)

    device = property(lambda s: s._device)
    tc_float = property(lambda s: torch.float32)

"""Harness that imports the *reference* Python (rsamf/add-gym, mounted read-only at
/root/reference) in the build container so that golden vectors can be generated from it.

TEST TOOLING ONLY.  Nothing here is imported by the product package, by bench.py's timed
path or by the GPU tests: the reference cannot travel to the GPU box, only the small .npz
fixtures written by tools/gen_golden.py do.

What is supplied here (all written for this repo, nothing copied):
  * empty stand-in modules for third-party imports the reference makes but that are not
    installed offline (tensorboard, genesis, mujoco, warp, mujoco_warp, hydra, omegaconf,
    torchvision, pyglet).  None of them is on the algorithmic path.
  * FakeEngine: a deterministic kinematic simulator implementing the reference's engine
    plugin API (add_gym/engine/base_engine.py:93-510) in plain torch on CPU.  Genesis itself
    is not installable offline, so physics parity is unpinned; every
    golden is taken downstream of the engine boundary.
"""
from __future__ import annotations

import os
import shutil
import sys
import tempfile
import types
import xml.etree.ElementTree as ET

import numpy as np
import torch

REF_ROOT = os.environ.get("ADD_GYM_REFERENCE", "/root/reference")


def install_stubs():
    """Register empty stand-ins for absent third-party packages (process-local)."""

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_image(self, *a, **k):
            pass

        def flush(self):
            pass

    if "tensorboard" not in sys.modules:
        mod("tensorboard")
    tb = mod("torch.utils.tensorboard", SummaryWriter=_SummaryWriter)
    import torch.utils as tu

    tu.tensorboard = tb
    mod("genesis")
    mod("mujoco")
    mod("warp")
    mod("mujoco_warp")
    mod("pyglet", options={})
    hyd = mod("hydra")
    hyd.main = lambda *a, **k: (lambda f: f)
    hu = mod("hydra.utils", instantiate=lambda cfg: FakeEngine(**{k: v for k, v in cfg.items() if k != "_target_"}))
    hyd.utils = hu
    mod("omegaconf", DictConfig=dict, OmegaConf=object)
    tv = mod("torchvision")
    tvt = mod("torchvision.transforms")
    tvf = mod("torchvision.transforms.functional", to_tensor=lambda x: x)
    tv.transforms = tvt
    tvt.functional = tvf
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    sys.dont_write_bytecode = True


# --------------------------------------------------------------------------------------
# Fake kinematic engine (reference plugin API, CPU torch)
# --------------------------------------------------------------------------------------
class FakeLink:
    def __init__(self, name, idx, idx_local):
        self._name, self._idx, self._idx_local = name, idx, idx_local

    idx = property(lambda s: s._idx)
    idx_local = property(lambda s: s._idx_local)
    name = property(lambda s: s._name)


class FakeJoint:
    def __init__(self, name, dofs_idx, limits):
        self._name, self._dofs, self._limits = name, list(dofs_idx), list(limits)

    dofs_idx = property(lambda s: s._dofs)
    dofs_idx_local = property(lambda s: s._dofs)
    dofs_limit = property(lambda s: s._limits)
    name = property(lambda s: s._name)


def parse_mjcf_bfs(path):
    """Bodies in breadth-first order (the order Genesis assigns links/dofs, see
    kin_char_model.py:117 comment) with their hinge joint name/range."""
    root = ET.parse(path).getroot()
    body0 = root.find("worldbody").find("body")
    out = []
    queue = [body0]
    while queue:
        b = queue.pop(0)
        j = b.find("joint")
        rng = None
        jname = None
        if j is not None and j.attrib.get("type", "hinge") != "free":
            jname = j.attrib["name"]
            rng = tuple(float(v) for v in j.attrib["range"].split())
        out.append((b.attrib["name"], jname, rng))
        queue.extend(b.findall("body"))
    return out


class FakeEntity:
    """Kinematic G1: joints follow their PD target with a first-order lag; the root keeps
    whatever set_qpos / set_dofs_velocity last wrote."""

    LAG = 0.5

    def __init__(self, scene, morph_file, link_base):
        self._scene = scene
        bodies = parse_mjcf_bfs(morph_file)
        self._links = [FakeLink(n, link_base + i, i) for i, (n, _, _) in enumerate(bodies)]
        inf = float("inf")
        self._joints = [FakeJoint("floating_base_joint", range(6), [(-inf, inf)] * 6)]
        d = 6
        for _, jn, rng in bodies[1:]:
            self._joints.append(FakeJoint(jn, [d], [rng]))
            d += 1
        self._n_dofs = d
        self.forced_contact_link = None  # [N] long (global link idx) or -1

    def build(self, n):
        self.n = n
        f = torch.float32
        self.pos = torch.zeros(n, 3, dtype=f)
        self.pos[:, 2] = 0.793
        self.quat = torch.zeros(n, 4, dtype=f)
        self.quat[:, 0] = 1
        self.vel = torch.zeros(n, 3, dtype=f)
        self.ang = torch.zeros(n, 3, dtype=f)
        self.dofs_pos = torch.zeros(n, self._n_dofs, dtype=f)
        self.dofs_vel = torch.zeros(n, self._n_dofs, dtype=f)
        self.target = torch.zeros(n, self._n_dofs - 6, dtype=f)
        self.forced_contact_link = torch.full((n,), -1, dtype=torch.long)

    # state getters (views, like Genesis)
    def get_pos(self):
        return self.pos

    def get_quat(self):
        return self.quat

    def get_vel(self):
        return self.vel

    def get_ang(self):
        return self.ang

    def get_dofs_position(self):
        return self.dofs_pos

    def get_dofs_velocity(self):
        return self.dofs_vel

    def _ids(self, envs_idx):
        return slice(None) if envs_idx is None else envs_idx

    def set_pos(self, pos, envs_idx=None):
        self.pos[self._ids(envs_idx)] = pos

    def set_quat(self, quat, envs_idx=None):
        self.quat[self._ids(envs_idx)] = quat

    def set_dofs_position(self, position, envs_idx=None, dofs_idx_local=None):
        ids = self._ids(envs_idx)
        if dofs_idx_local is None:
            self.dofs_pos[ids] = position
        else:
            tmp = self.dofs_pos[ids]
            tmp[:, dofs_idx_local] = position
            self.dofs_pos[ids] = tmp

    def set_dofs_velocity(self, velocity, envs_idx=None):
        ids = self._ids(envs_idx)
        self.dofs_vel[ids] = velocity
        self.vel[ids] = velocity[:, 0:3]
        self.ang[ids] = velocity[:, 3:6]

    def set_qpos(self, qpos, envs_idx=None):
        ids = self._ids(envs_idx)
        self.pos[ids] = qpos[:, 0:3]
        self.quat[ids] = qpos[:, 3:7]
        tmp = self.dofs_pos[ids]
        tmp[:, 6:] = qpos[:, 7:]
        self.dofs_pos[ids] = tmp
        t = self.target[ids]
        t[:] = qpos[:, 7:]
        self.target[ids] = t

    def control_dofs_position(self, position, dofs_idx_local=None):
        self.target[:] = position

    def set_dofs_kp(self, kp):
        self.kp = kp

    def set_dofs_kv(self, kv):
        self.kv = kv

    def zero_all_dofs_velocity(self, envs_idx=None):
        ids = self._ids(envs_idx)
        self.dofs_vel[ids] = 0
        self.vel[ids] = 0
        self.ang[ids] = 0

    def get_links_pos(self):
        return self.pos[:, None, :].repeat(1, len(self._links), 1)

    def get_links_quat(self):
        return self.quat[:, None, :].repeat(1, len(self._links), 1)

    def get_links_net_contact_force(self):
        return torch.zeros(self.n, len(self._links), 3)

    def get_contacts(self, with_entity=None, exclude_self_contact=False):
        la = self.forced_contact_link[:, None].clone()
        valid = la >= 0
        return {"link_a": la.clamp(min=0), "link_b": torch.zeros_like(la), "valid_mask": valid}

    def get_AABB(self):
        box = torch.zeros(self.n, 2, 3)
        box[:, 0] = torch.tensor([-0.2, -0.3, 0.0])
        box[:, 1] = torch.tensor([0.2, 0.3, 1.3])
        return box

    def get_joint(self, name):
        return next(j for j in self._joints if j.name == name)

    def get_link(self, name):
        return next(l for l in self._links if l.name == name)

    joints = property(lambda s: s._joints)
    links = property(lambda s: s._links)
    n_dofs = property(lambda s: s._n_dofs)

    def step(self, dt):
        q = self.dofs_pos[:, 6:]
        qn = q + self.LAG * (self.target - q)
        self.dofs_vel[:, 6:] = (qn - q) / dt
        self.dofs_pos[:, 6:] = qn


class FakePlane:
    def __init__(self):
        self._links = [FakeLink("plane", 0, 0)]

    links = property(lambda s: s._links)


class FakeCamera:
    _in_recording = False

    def follow_entity(self, e):
        pass

    def start_recording(self):
        pass

    def stop_recording(self, filename, fps=30):
        pass

    def render(self):
        pass


class FakeScene:
    def __init__(self, dt):
        self._dt = dt
        self._t = 0
        self._entities = []
        self._n_links = 0

    def add_entity(self, morph_type, morph_file=None, morph_pos=None, morph_quat=None,
                   material_type=None, visualize_contact=True):
        if morph_type == "plane":
            e = FakePlane()
            self._n_links += 1
            return e
        e = FakeEntity(self, morph_file, self._n_links)
        self._n_links += len(e.links)
        self._entities.append(e)
        return e

    def add_camera(self, **k):
        return FakeCamera()

    def build(self, n_envs, env_spacing=None):
        for e in self._entities:
            e.build(n_envs)

    def step(self):
        for e in self._entities:
            e.step(self._dt)
        self._t += 1

    t = property(lambda s: s._t)


class FakeEngine:
    def __init__(self, **cfg):
        self.cfg = cfg

    def init(self, backend, precision):
        pass

    def create_scene(self, show_viewer, sim_options, rigid_options, vis_options=None, viewer_options=None):
        return FakeScene(sim_options["dt"])

    device = property(lambda s: torch.device("cpu"))
    tc_float = property(lambda s: torch.float32)


# --------------------------------------------------------------------------------------
# config + scratch helpers
# --------------------------------------------------------------------------------------
def load_ref_config(num_envs, motion_file, **agent_overrides):
    """The reference's Hydra tree composed by hand from its YAML files (train.yaml defaults
    list, configs/train.yaml:2-8)."""
    import yaml

    cdir = os.path.join(REF_ROOT, "add_gym", "configs")

    def y(p):
        with open(os.path.join(cdir, p)) as f:
            return yaml.safe_load(f)

    cfg = {
        "agent": y("agent/add_g1.yaml"),
        "engine": y("engine/genesis.yaml"),
        "robot": y("robot/g1.yaml"),
        "task": y("task/pose.yaml"),
        "distributed": y("distributed/ddp.yaml"),
        "mode": "train",
        "experiment_name": "golden",
        "log_dir": tempfile.mkdtemp(prefix="addgym_logs_"),
    }
    cfg["engine"].update(num_envs=num_envs, enable_viewer=False, enable_video_recording=False)
    cfg["robot"]["urdf_path"] = os.path.join(REF_ROOT, "assets/g1_description/g1_29.xml")
    cfg["task"]["motion_file"] = motion_file
    cfg["agent"].update(agent_overrides)
    return cfg


def scratch_clip(name, max_frames=None, dst_dir=None):
    """Copy (a prefix of) a reference clip to a writable dir: load_motion() writes a .pkl next
    to the file it reads (anim/motion.py:41-42) and the reference mount is read-only."""
    dst_dir = dst_dir or tempfile.mkdtemp(prefix="addgym_clips_")
    src = os.path.join(REF_ROOT, "assets/motions", name)
    dst = os.path.join(dst_dir, name)
    if max_frames is None:
        shutil.copy(src, dst)
    else:
        with open(src) as f, open(dst, "w") as g:
            for i, line in enumerate(f):
                if i >= max_frames:
                    break
                g.write(line)
    return dst

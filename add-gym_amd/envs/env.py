"""Environment: engine (plugin, chosen by `engine._target_`), scene, ground plane, robot and the fp32 episode clock.
Same construction sequence and attributes as the reference's add_gym/envs/env.py:9-190 (viewer / video hooks are
out of scope and ignored)."""
import importlib

import torch

from ..robot import Manipulator


def instantiate(cfg):
    """Minimal hydra.utils.instantiate: import `_target_` and call it with the remaining keys (env.py:35)."""
    try:
        from hydra.utils import instantiate as hydra_instantiate  # real Hydra when available

        return hydra_instantiate(cfg)
    except ImportError:
        pass
    target = cfg["_target_"]
    mod, _, name = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), name)
    return cls(**{k: v for k, v in cfg.items() if k != "_target_"})


class Environment:
    def __init__(self, config, device):
        self.device = device
        self.env_cfg, self.engine_cfg, self.robot_cfg, self.task_cfg = config, config["engine"], config["robot"], config["task"]
        self.ctrl_dt = self.engine_cfg["ctrl_dt"]
        self.engine = instantiate(self.engine_cfg)
        self.engine.init(backend="gpu" if torch.cuda.is_available() else "cpu", precision="32")
        self.scene = self.engine.create_scene(
            show_viewer=False, sim_options={"dt": self.ctrl_dt},
            rigid_options={"dt": self.ctrl_dt, "constraint_solver": "Newton", "enable_collision": True, "enable_self_collision": True,
                           "enable_joint_limit": True}, vis_options=None, viewer_options=None)
        self.plane = self.scene.add_entity(morph_type="plane")
        self.robot = Manipulator(num_envs=self.engine_cfg["num_envs"], scene=self.scene, engine=self.engine, robot_cfg=self.robot_cfg,
                                 env_spacing=self.engine_cfg["env_spacing"], enable_ref=False, device=device)
        spacing = self.engine_cfg["env_spacing"]
        self.scene.build(n_envs=self.engine_cfg["num_envs"], env_spacing=(spacing, spacing))
        self.robot.on_build()
        self.num_envs = self.engine_cfg["num_envs"]
        self.time_buf = torch.zeros(self.num_envs, device=self.engine.device, dtype=torch.float32)  # env.py:124
        self.extras = {}

    def step(self, actions):
        """env.py:150-155.  (The HIP env step advances time_buf itself; this method serves callers that drive the
        environment without the fused kernel.)"""
        self.robot.apply_action(actions)
        self.scene.step()
        self.time_buf += self.ctrl_dt

    def reset_idx(self, envs_idx):
        if len(envs_idx) > 0:
            self.time_buf[envs_idx] = 0

    def reset(self, env_ids=None):
        self.reset_idx(torch.arange(self.num_envs, device=self.device) if env_ids is None else env_ids)


class ImitationEnvironment(Environment):
    def __init__(self, config, device):
        super().__init__(config, device)
        self._diagnostics = {}

    def get_reward_succ(self):
        return 0.0  # env.py:181-184

    def get_reward_fail(self):
        return 0.0

    def get_diagnostics(self):
        return self._diagnostics

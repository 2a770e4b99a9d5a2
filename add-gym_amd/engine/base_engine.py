"""Simulator plugin API -- the drop-in seam.  Same class and method names, arguments and shapes as
the reference's add_gym/engine/base_engine.py:13-510, so an engine written for add-gym plugs in
here unchanged (and KinematicEngine below plugs into add-gym).  Engines are chosen by the
`_target_` key of the `engine` config group (configs/engine/*.yaml, envs/env.py).

Shapes: N envs; quaternions wxyz; dofs 0..5 = floating base, 6.. = joints in breadth-first order.
"""
from abc import ABC, abstractmethod
from typing import Dict, List, Optional, Tuple

import torch


class BaseCamera(ABC):
    @abstractmethod
    def follow_entity(self, entity: "BaseEntity") -> None: ...
    @abstractmethod
    def start_recording(self) -> None: ...
    @abstractmethod
    def stop_recording(self, filename: str, fps: int = 30) -> None: ...
    @abstractmethod
    def render(self) -> None: ...
    @property
    @abstractmethod
    def _in_recording(self) -> bool: ...


class BaseLink(ABC):
    @property
    @abstractmethod
    def idx(self) -> int: ...        # index over all links of the scene
    @property
    @abstractmethod
    def idx_local(self) -> int: ...  # index inside its entity
    @property
    @abstractmethod
    def name(self) -> str: ...


class BaseJoint(ABC):
    @property
    @abstractmethod
    def dofs_idx(self) -> List[int]: ...
    @property
    @abstractmethod
    def dofs_idx_local(self) -> List[int]: ...
    @property
    @abstractmethod
    def dofs_limit(self) -> List[Tuple[float, float]]: ...
    @property
    @abstractmethod
    def name(self) -> str: ...


class BaseEntity(ABC):
    # root pose / velocity: [N,3], [N,4] wxyz, [N,3], [N,3]
    @abstractmethod
    def get_pos(self) -> torch.Tensor: ...
    @abstractmethod
    def set_pos(self, pos: torch.Tensor, envs_idx: Optional[torch.Tensor] = None) -> None: ...
    @abstractmethod
    def get_quat(self) -> torch.Tensor: ...
    @abstractmethod
    def set_quat(self, quat: torch.Tensor, envs_idx: Optional[torch.Tensor] = None) -> None: ...
    @abstractmethod
    def get_vel(self) -> torch.Tensor: ...
    @abstractmethod
    def get_ang(self) -> torch.Tensor: ...
    # dofs: [N,n_dofs]
    @abstractmethod
    def get_dofs_position(self) -> torch.Tensor: ...
    @abstractmethod
    def set_dofs_position(self, position: torch.Tensor, envs_idx: Optional[torch.Tensor] = None,
                          dofs_idx_local: Optional[List[int]] = None) -> None: ...
    @abstractmethod
    def get_dofs_velocity(self) -> torch.Tensor: ...
    @abstractmethod
    def set_dofs_velocity(self, velocity: torch.Tensor, envs_idx: Optional[torch.Tensor] = None) -> None: ...
    @abstractmethod
    def control_dofs_position(self, position: torch.Tensor, dofs_idx_local: Optional[List[int]] = None) -> None: ...
    @abstractmethod
    def set_dofs_kp(self, kp: torch.Tensor) -> None: ...
    @abstractmethod
    def set_dofs_kv(self, kv: torch.Tensor) -> None: ...
    @abstractmethod
    def zero_all_dofs_velocity(self, envs_idx: Optional[torch.Tensor] = None) -> None: ...
    # links
    @abstractmethod
    def get_links_pos(self) -> torch.Tensor: ...
    @abstractmethod
    def get_links_quat(self) -> torch.Tensor: ...
    @abstractmethod
    def get_links_net_contact_force(self) -> torch.Tensor: ...
    @abstractmethod
    def get_contacts(self, with_entity: Optional["BaseEntity"] = None, exclude_self_contact: bool = False) -> Dict[str, torch.Tensor]:
        """{'link_a','link_b': [N,C] int64 global link ids, 'valid_mask': [N,C] bool}"""
    @abstractmethod
    def get_AABB(self) -> torch.Tensor: ...
    @abstractmethod
    def get_joint(self, name: str) -> BaseJoint: ...
    @abstractmethod
    def get_link(self, name: str) -> BaseLink: ...
    @property
    @abstractmethod
    def joints(self) -> List[BaseJoint]: ...
    @property
    @abstractmethod
    def links(self) -> List[BaseLink]: ...
    @property
    @abstractmethod
    def n_dofs(self) -> int: ...
    @abstractmethod
    def set_qpos(self, qpos: torch.Tensor, envs_idx: Optional[torch.Tensor] = None) -> None:
        """qpos [n,36] = root xyz, root quat wxyz, 29 joint angles"""


class BaseScene(ABC):
    @abstractmethod
    def add_entity(self, morph_type: str, morph_file: Optional[str] = None, morph_pos=(0.0, 0.0, 0.0),
                   morph_quat=(1.0, 0.0, 0.0, 0.0), material_type: str = "rigid", visualize_contact: bool = False) -> BaseEntity: ...
    @abstractmethod
    def add_camera(self, res=(640, 480), pos=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, 0.0), fov: float = 40) -> BaseCamera: ...
    @abstractmethod
    def build(self, n_envs: int, env_spacing: Tuple[float, float]) -> None: ...
    @abstractmethod
    def step(self) -> None: ...
    @property
    @abstractmethod
    def t(self) -> int: ...


class BaseEngine(ABC):
    @abstractmethod
    def init(self, backend: str, precision: str) -> None: ...
    @abstractmethod
    def create_scene(self, show_viewer: bool, sim_options: dict, rigid_options: dict, vis_options: Optional[dict] = None,
                     viewer_options: Optional[dict] = None) -> BaseScene: ...
    @property
    @abstractmethod
    def device(self) -> torch.device: ...
    @property
    @abstractmethod
    def tc_float(self) -> torch.dtype: ...

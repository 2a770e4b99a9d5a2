"""Kinematic playback export (the GUI-less counterpart of the reference's view.py): runs the deterministic policy (or, with
`view.source=reference`, just the clip) for `view.steps` control steps and writes the trajectories an external viewer needs
as one .npz: per step and env the simulator's root position / wxyz quaternion / joint angles, the tracked reference
pose, reward, done flag, plus the kinematic tree's body names, parents and hinge axes.

    python -m add_gym_amd.view mode=test engine.num_envs=4 resume_path=output/<run>/model.pt view.out=playback.npz view.steps=600
"""
import sys

import numpy as np
import torch

from . import _lib as L


def export_playback(cfg, out_file, steps, source="policy"):
    from .learning.add_agent import ADDAgent, AgentMode

    agent = ADDAgent(cfg)
    if cfg.get("resume_path"):
        agent.load(cfg["resume_path"])
    agent.set_mode(AgentMode.TEST)
    agent.reset_all_envs()
    lib, S, N = agent._motion_lib, agent._S, agent.N
    rec = {k: [] for k in ("sim_pose", "sim_vel", "ref_pose", "ref_vel", "reward", "done", "motion_id", "motion_time", "time")}
    reward = torch.zeros(N, device=agent._device)
    out = L.StepOutT(L.ptr(agent._B["obs"][0]), None, None, L.ptr(agent._B["disc_obs"][agent.T]), L.ptr(agent._B["disc_demo"][agent.T]), L.ptr(reward),
                     None, None, None, None)
    for k in range(steps):
        if source == "reference":  # no policy, no engine: the character is placed on the clip at the time the step arrives at
            t_next = (S["time"] + agent._env.ctrl_dt) + S["time_off"]  # same fp32 association as the step kernel
            pose, vel = lib.get_precomputed_motion_step(S["motion_id"], t_next, packed=True)
            S["sim_pose"].copy_(pose)
            S["sim_vel"].copy_(vel)
            L.call("addhip_env_step", lib.c_struct, agent._task, agent._env_c_test, out, agent._head, agent._stream())
            agent._head = (agent._head + 1) % agent._task.num_disc_obs_steps
        else:
            agent._decide_action(0, 0, True)
            agent._step_env(0, out, agent._env_c_test)
        t = S["time"] + S["time_off"]
        ref_pose, ref_vel = lib.get_precomputed_motion_step(S["motion_id"], t, packed=True)
        for key, val in (("sim_pose", S["sim_pose"]), ("sim_vel", S["sim_vel"]), ("ref_pose", ref_pose), ("ref_vel", ref_vel), ("reward", reward),
                         ("done", S["done"]), ("motion_id", S["motion_id"]), ("motion_time", t), ("time", S["time"])):
            rec[key].append(val.detach().cpu().numpy().copy())
        agent._reset_envs(False, agent._B["obs"][0], agent._B["disc_obs"][agent.T], agent._B["disc_demo"][agent.T], (7 << 40) + k)  # Philox namespace 7: playback resets
    kin = agent._env.robot._kin_char_model
    arrays = {k: np.stack(v) for k, v in rec.items()}
    np.savez_compressed(out_file, dt=np.float32(agent._env.ctrl_dt), layout="pose = root xyz | root quaternion wxyz | 29 joint angles (kinematic-tree order)",
                        body_names=np.array(kin.get_body_names()), parents=np.asarray(kin._parents, np.int32), joint_axes=np.asarray(kin.joint_axes(), np.float32),
                        **arrays)
    return arrays


def main(argv=None):
    from .config import load_config

    argv = list(sys.argv[1:] if argv is None else argv)
    opts = {"view.out": "playback.npz", "view.steps": "300", "view.source": "policy"}
    for a in list(argv):
        k = a.split("=", 1)[0]
        if k in opts:
            opts[k] = a.split("=", 1)[1]
            argv.remove(a)
    cfg = load_config("test", argv)
    arrays = export_playback(cfg, opts["view.out"], int(opts["view.steps"]), opts["view.source"])
    print(f"wrote {opts['view.out']}: {arrays['sim_pose'].shape[0]} steps x {arrays['sim_pose'].shape[1]} envs, mean reward {arrays['reward'].mean():.4f}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python (imported from
/root/reference on CPU in the build container; see tools/ref_harness.py).

Run:  python tools/gen_golden.py [names...]      (from the repo root, build container only)

Fixtures are data only: seeded inputs, the random draws the reference made, and the outputs
it produced.  The reference's source never enters the repo.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness as H  # noqa: E402

H.install_stubs()
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
XML = os.path.join(H.REF_ROOT, "assets/g1_description/g1_29.xml")


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path}  {os.path.getsize(path)/1024:.1f} KiB")


def T(x, dtype=torch.float32):
    return torch.tensor(np.asarray(x), dtype=dtype)


def rand_quats(rng, n):
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    return q


# ------------------------------------------------------------------------------------
def gen_quat_math():
    import add_gym.util.torch_util as tu

    rng = np.random.RandomState(11)
    n = 256
    q0, q1 = rand_quats(rng, n), rand_quats(rng, n)
    # near-degenerate cases: identity-ish, equal pairs, antipodal pairs, tiny rotations
    q0[0] = [1, 0, 0, 0]
    q1[0] = [1, 0, 0, 0]
    q0[1] = [-1, 0, 0, 0]
    q1[2] = q0[2]
    q1[3] = -q0[3]
    eps = rng.standard_normal((8, 4)).astype(np.float32) * 1e-4
    q1[4:12] = q0[4:12] + eps
    q1[4:12] /= np.linalg.norm(q1[4:12], axis=-1, keepdims=True)
    q0[12:16, 1:] *= 1e-7
    q0[12:16] /= np.linalg.norm(q0[12:16], axis=-1, keepdims=True)
    v = rng.standard_normal((n, 3)).astype(np.float32)
    t = rng.rand(n).astype(np.float32)
    t[:4] = [0, 1, 0.5, 0.25]
    axis = rng.standard_normal((n, 3)).astype(np.float32)
    unit_axis = np.eye(3, dtype=np.float32)[rng.randint(0, 3, n)]
    angle = (rng.rand(n).astype(np.float32) * 2 - 1) * 3.0
    a0, a1, tv, tt, ta, tua, tang = map(T, (q0, q1, v, t, axis, unit_axis, angle))
    ax, ang = tu.quat_to_axis_angle(a0)
    save(
        "quat_math",
        q0=q0, q1=q1, v=v, t=t, axis=axis, unit_axis=unit_axis, angle=angle,
        quat_mul=tu.quat_mul(a0, a1), quat_rotate=tu.quat_rotate(a0, tv),
        quat_pos=tu.quat_pos(a0), quat_conjugate=tu.quat_conjugate(a0),
        axis_angle_axis=ax, axis_angle_angle=ang,
        quat_to_exp_map=tu.quat_to_exp_map(a0),
        quat_to_tan_norm=tu.quat_to_tan_norm(a0),
        quat_diff_angle=tu.quat_diff_angle(a0, a1),
        slerp=tu.slerp(a0, a1, tt),
        calc_heading=tu.calc_heading(a0),
        calc_heading_quat_inv=tu.calc_heading_quat_inv(a0),
        axis_angle_to_quat=tu.axis_angle_to_quat(ta, tang),
        quat_twist_angle=tu.quat_twist_angle(a0, tua),
        quat_normalize=tu.quat_normalize(a0 * 1.7),
    )


def ref_kin():
    import add_gym.anim.kin_char_model as kcm

    m = kcm.KinCharModel("cpu")
    m.load_char_file(XML)
    return m


def task_cfg():
    import yaml

    with open(os.path.join(H.REF_ROOT, "add_gym/configs/task/pose.yaml")) as f:
        return yaml.safe_load(f)


def ref_motion_lib(motion_file):
    import add_gym.anim.motion_lib as ml

    return ml.MotionLib(motion_file, list(task_cfg()["motion_joint_order"]), ref_kin(), 0.01, "cpu")


def lib_tables(lib):
    return dict(
        step_root_pos=lib._step_root_pos, step_root_rot=lib._step_root_rot, step_root_vel=lib._step_root_vel,
        step_root_ang_vel=lib._step_root_ang_vel, step_dof_pos=lib._step_dof_pos, step_dof_vel=lib._step_dof_vel,
        lengths=lib._motion_lengths, num_frames=lib._motion_num_frames, start_idx=lib._motion_start_idx,
        weights=lib._motion_weights, loop_modes=lib._motion_loop_modes,
    )


def two_clip_yaml(n0=60, n1=45):
    d = os.path.dirname(H.scratch_clip("walk1_subject1_trimmed.motion", n0))
    c0 = os.path.join(d, "walk1_subject1_trimmed.motion")
    c1 = H.scratch_clip("run1_subject2.motion", n1, dst_dir=d)
    y = os.path.join(d, "two.yaml")
    with open(y, "w") as f:
        f.write(f"motions:\n  - file: {c0}\n    weight: 1.0\n  - file: {c1}\n    weight: 3.0\n")
    return y, c0, c1


def gen_kin_tree():
    m = ref_kin()
    axes = torch.stack([j.axis if j.axis is not None else torch.zeros(3) for j in m._joints])
    # action space needs the robot wrapper -> build the env through the fake engine
    cfg = H.load_ref_config(4, H.scratch_clip("walk1_subject1_trimmed.motion", 40))
    from add_gym.envs.env import ImitationEnvironment

    env = ImitationEnvironment(cfg, "cpu")
    a_space = env.robot.get_action_space()
    order = list(task_cfg()["motion_joint_order"])
    save(
        "kin_tree",
        meta=json.dumps(dict(body_names=m.get_body_names(), joint_names=m.get_joint_order(), motion_joint_order=order)),
        parents=m._parent_indices, axes=axes, action_space=a_space,
        motion_idx=np.asarray([order.index(n) for n in m.get_joint_order()[1:]]),
    )


def gen_motion_small():
    from add_gym.anim import motion as ref_motion

    clip = H.scratch_clip("walk1_subject1_trimmed.motion", 200)
    frames = ref_motion.load_motion(clip).frames
    lib = ref_motion_lib(clip)
    y, c0, c1 = two_clip_yaml()
    lib2 = ref_motion_lib(y)
    f0 = ref_motion.load_motion(c0).frames
    f1 = ref_motion.load_motion(c1).frames
    two = {"two_" + k: v for k, v in lib_tables(lib2).items()}
    save("motion_small", frames=frames, two_frames0=f0, two_frames1=f1, **lib_tables(lib), **two)


def gen_lookup():
    clip = H.scratch_clip("walk1_subject1_trimmed.motion", 200)
    lib = ref_motion_lib(clip)
    y, _, _ = two_clip_yaml()
    lib2 = ref_motion_lib(y)
    rng = np.random.RandomState(5)

    def idx_of(L, ids, times):
        # motion_lib.py:322-326 evaluated by the reference's own tensors
        mf = (times * L._dt_inv).long()
        mf = torch.clip(mf, 0, L._step_root_pos.shape[-2] - 1)
        return mf + L._motion_start_idx[ids]

    # accumulated fp32 clock: time_buf += 0.01 (env.py:155)
    clock = torch.zeros(1, dtype=torch.float32)
    acc = []
    for _ in range(2000):
        clock += 0.01
        acc.append(clock.clone())
    acc = torch.cat(acc)
    off = T(rng.rand(2000).astype(np.float32) * 0.5)
    off = (off // 0.01) * 0.01
    times = torch.cat([acc + off, T(rng.rand(500).astype(np.float32) * 8.0 - 0.5)])
    ids = torch.zeros(times.shape[0], dtype=torch.long)
    rp, rr, rv, ra, dp, dv = lib.get_precomputed_motion_step(ids, times)
    ids2 = T(rng.randint(0, 2, 600), torch.long)
    times2 = T(rng.rand(600).astype(np.float32) * 1.2)  # short clips: stays inside the table
    idx2 = idx_of(lib2, ids2, times2)
    ok = idx2 < lib2._step_root_pos.shape[0]
    ids2, times2, idx2 = ids2[ok], times2[ok], idx2[ok]
    rp2 = lib2.get_precomputed_motion_step(ids2, times2)[0]
    save("lookup", acc_clock=acc, times=times, ids=ids, idx=idx_of(lib, ids, times), root_pos=rp, dof_vel=dv,
         two_ids=ids2, two_times=times2, two_idx=idx2, two_root_pos=rp2)


GENS = dict(quat_math=gen_quat_math, kin_tree=gen_kin_tree, motion_small=gen_motion_small, lookup=gen_lookup)

if __name__ == "__main__":
    from gen_golden_agent import AGENT_GENS  # noqa: E402

    GENS.update(AGENT_GENS)
    names = sys.argv[1:] or list(GENS)
    torch.manual_seed(0)
    for n in names:
        print("==", n)
        GENS[n]()

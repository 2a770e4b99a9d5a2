"""Rigid-body engine on the GPU (csrc/rigid.hip through the C ABI, engine/rigid_engine.py) against its float64 oracle
(oracle/rigid.py) and against physics: one control step on 64 random states with and without ground contact, free flight
(momentum / centre-of-mass closed form), standing, the BaseEntity plugin surface, and a training iteration on it.
Physics parity with the reference's simulators (Genesis / MuJoCo-Warp) is UNPINNED: neither is installable, no fixtures exist."""
import copy

import numpy as np
import pytest

from oracle import rigid as RB
from tests.util import G1_XML, kin_meta

pytestmark = pytest.mark.gpu
F = np.float32


def make_entity(n, **opts):
    import torch
    import add_gym_amd  # noqa: F401
    from add_gym_amd.engine.rigid_engine import RigidBodyEngine

    eng = RigidBodyEngine(**opts)
    eng.init("gpu", "32")
    scene = eng.create_scene(False, {"dt": 0.01}, {})
    plane = scene.add_entity("plane")
    ent = scene.add_entity("mjcf", G1_XML)
    scene.build(n)
    m = RB.RigidModel(G1_XML)
    kp, kv = RB.gains(m)
    ent.set_dofs_kp(torch.cat([torch.zeros(6), torch.tensor(kp, dtype=torch.float32)]))
    ent.set_dofs_kv(torch.cat([torch.zeros(6), torch.tensor(kv, dtype=torch.float32)]))
    return eng, scene, plane, ent, m, kp, kv


def rand_states(rng, n, zlo, zhi):
    q = rng.uniform(-0.4, 0.4, (n, 29))
    quat = rng.standard_normal((n, 4)) * 0.3 + np.array([1.0, 0, 0, 0])
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    return RB.State(np.c_[rng.uniform(-2, 2, (n, 2)), rng.uniform(zlo, zhi, n)], quat, q, rng.standard_normal((n, 3)) * 0.5, rng.standard_normal((n, 3)),
                    rng.standard_normal((n, 29)) * 2)


def put(ent, st):
    import torch

    pose, vel = st.packed()
    ent.pose.copy_(torch.tensor(pose.astype(F)))
    ent.vel.copy_(torch.tensor(vel.astype(F)))


def get(ent):
    return RB.State.from_packed(ent.pose.cpu().numpy().astype(np.float64), ent.vel.cpu().numpy().astype(np.float64))


@pytest.mark.parametrize("lanes", [4, 1])
@pytest.mark.parametrize("case", ["contact", "flight"])
def test_one_control_step_matches_the_float64_oracle(case, lanes):
    """Both kernels: four lanes per env (one per chain of the tree; the default) and one lane per env; 70 envs = ragged last
    workgroup of either."""
    import torch

    n = 70
    eng, scene, plane, ent, m, kp, kv = make_entity(n, lanes_per_env=lanes)
    assert (ent._d_chains is not None) == (lanes == 4)
    rng = np.random.RandomState(3 if case == "contact" else 4)
    st = rand_states(rng, n, 0.25, 0.85) if case == "contact" else rand_states(rng, n, 2.0, 3.0)
    # fp32 inputs on both sides
    pose, vel = (a.astype(F).astype(np.float64) for a in st.packed())
    st = RB.State.from_packed(pose, vel)
    tgt = rng.uniform(-0.5, 0.5, (n, 29)).astype(F)
    put(ent, st)
    ent.control_dofs_position(torch.tensor(tgt, device="cuda"))
    scene.step()
    torch.cuda.synchronize()
    got = get(ent)
    want, touch = RB.step(m, RB.RigidParams(), kp, kv, st, tgt.astype(np.float64))
    if case == "contact":
        assert touch.sum() > 100
    # tolerance 1e-5 of each quantity's scale over the batch (fp32 kernel vs float64 oracle, 4 substeps incl. stiff contacts)
    for name in ("root_pos", "root_quat", "q", "root_vel", "root_ang", "qd"):
        a, b = getattr(got, name), getattr(want, name)
        scale = max(1.0, np.abs(b).max())
        assert np.abs(a - b).max() <= 1e-5 * scale * (10 if case == "contact" else 1), (name, float(np.abs(a - b).max()), float(scale))
    bits = ent.contact_bits.cpu().numpy().astype(np.uint32)
    want_bits = (touch.astype(np.uint32) << np.arange(m.nb, dtype=np.uint32)).sum(1)
    assert (bits != want_bits).sum() <= 1  # (a sphere within rounding of the ground may flip)
    c = ent.get_contacts(with_entity=plane)
    assert c["link_a"].shape == (n, m.nb) and c["valid_mask"].dtype == torch.bool and c["link_b"].eq(plane.links[0].idx).all()
    assert np.array_equal(c["valid_mask"].cpu().numpy(), ((bits[:, None] >> np.arange(m.nb)) & 1).astype(bool))


@pytest.mark.parametrize("case", ["contact", "flight"])
def test_register_form_of_the_four_lane_kernel_matches_the_float64_oracle(case):
    """Launches of more than two waves per CU take the form of the four-lane kernel that keeps the per-body pass state in step-indexed
    VGPR arrays (38 KB of LDS per workgroup: four waves per CU).  The 70 oracle states, tiled over enough envs to select it: the first 70
    against the float64 oracle at the usual tolerance, every replica bit-identical to its original (a ragged last workgroup included)."""
    import torch

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    reps = (2 * 16 * cus) // 70 + 2
    n = 70 * reps - 3
    eng, scene, plane, ent, m, kp, kv = make_entity(n, lanes_per_env=4)
    assert n > 2 * 16 * cus
    rng = np.random.RandomState(3 if case == "contact" else 4)
    st = rand_states(rng, 70, 0.25, 0.85) if case == "contact" else rand_states(rng, 70, 2.0, 3.0)
    pose, vel = (a.astype(F) for a in st.packed())
    tgt = rng.uniform(-0.5, 0.5, (70, 29)).astype(F)
    tile = lambda a: torch.tensor(np.tile(a, (reps, 1))[:n], device="cuda")
    ent.pose.copy_(tile(pose))
    ent.vel.copy_(tile(vel))
    ent.control_dofs_position(tile(tgt))
    scene.step()
    torch.cuda.synchronize()
    got = get(ent)
    st64 = RB.State.from_packed(pose.astype(np.float64), vel.astype(np.float64))
    want, touch = RB.step(m, RB.RigidParams(), kp, kv, st64, tgt.astype(np.float64))
    for name in ("root_pos", "root_quat", "q", "root_vel", "root_ang", "qd"):
        a, b = getattr(got, name)[:70], getattr(want, name)
        scale = max(1.0, np.abs(b).max())
        assert np.abs(a - b).max() <= 1e-5 * scale * (10 if case == "contact" else 1), (name, float(np.abs(a - b).max()), float(scale))
    P, V, Bt = ent.pose.cpu().numpy(), ent.vel.cpu().numpy(), ent.contact_bits.cpu().numpy()
    for r in range(1, reps):
        k = min(70, n - 70 * r)
        assert np.array_equal(P[70 * r:70 * r + k], P[:k]) and np.array_equal(V[70 * r:70 * r + k], V[:k]) and np.array_equal(Bt[70 * r:70 * r + k], Bt[:k]), r


def test_free_flight_follows_the_centre_of_mass_and_keeps_momentum():
    import torch

    n = 32
    eng, scene, plane, ent, m, kp, kv = make_entity(n, substeps=10, limit_stiffness=0.0)
    m0 = copy.deepcopy(m)
    ent.set_dofs_kp(torch.zeros(35))
    ent.set_dofs_kv(torch.zeros(35))
    rng = np.random.RandomState(5)
    st0 = rand_states(rng, n, 4.0, 5.0)
    put(ent, st0)
    st0 = get(ent)
    P0, L0, ke0, pe0, com0 = RB.momentum_and_energy(m0, st0)
    ent.control_dofs_position(torch.zeros(n, 29, device="cuda"))
    steps = 20
    for _ in range(steps):
        scene.step()
    torch.cuda.synchronize()
    s = get(ent)
    P, L, ke, pe, com = RB.momentum_and_energy(m0, s)
    T = steps * 0.01
    g = np.array([0, 0, -m.total_mass * RB.GRAVITY * T])
    assert np.abs(P - (P0 + g)).max() < 3e-3 * m.total_mass * RB.GRAVITY * T          # h = 1 ms: first-order integrator error
    assert np.abs(com[:, 2] - (com0[:, 2] + P0[:, 2] / m.total_mass * T - 0.5 * RB.GRAVITY * T * T)).max() < 2e-3
    assert np.abs(com[:, :2] - (com0[:, :2] + P0[:, :2] / m.total_mass * T)).max() < 2e-3
    Lc0, Lc = L0 - np.cross(com0, P0), L - np.cross(com, P)
    assert np.abs(Lc - Lc0).max() < 1e-2 * np.abs(Lc0).max()
    assert int(ent.contact_bits.abs().sum()) == 0


def test_robot_stands_on_its_feet_and_falls_when_dropped_sideways():
    import torch

    n = 8
    eng, scene, plane, ent, m, kp, kv = make_entity(n, max_torque=1e9)
    k6, v6 = RB.gains(m, 6.0)
    ent.set_dofs_kp(torch.cat([torch.zeros(6), torch.tensor(k6, dtype=torch.float32)]))
    ent.set_dofs_kv(torch.cat([torch.zeros(6), torch.tensor(v6, dtype=torch.float32)]))
    ent.set_termination_links(kin_meta().get("contact_bodies", ["left_knee_link", "left_ankle_pitch_link", "left_ankle_roll_link", "right_knee_link",
                                                                "right_ankle_pitch_link", "right_ankle_roll_link"]))
    st = RB.State(np.tile([0, 0, 0.8], (n, 1)), np.tile([1.0, 0, 0, 0], (n, 1)), np.zeros((n, 29)), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 29)))
    R, p, _ = RB.forward_kinematics(m, st.root_pos, st.root_quat, st.q)
    zmin = min(p[0, b, 2] + R[0, b, 2, :] @ r - rad for b, r, rad in zip(m.pt_body, m.pt_pos, m.pt_rad))
    st.root_pos[:, 2] -= zmin - 0.0005
    z0 = st.root_pos[0, 2]
    # envs 4..7 start lying on their side half a metre up: they must end on the ground with non-foot contacts
    st.root_quat[4:] = [np.sqrt(0.5), np.sqrt(0.5), 0, 0]
    st.root_pos[4:, 2] = 0.5
    put(ent, st)
    ent.control_dofs_position(torch.zeros(n, 29, device="cuda"))
    aabb = ent.get_AABB().cpu().numpy()
    assert abs(aabb[0, 0, 2] - 0.0005) < 1e-3 and 1.2 < aabb[0, 1, 2] < 1.4  # feet on the ground, about 1.3 m tall
    for _ in range(150):
        scene.step()
    torch.cuda.synchronize()
    s = get(ent)
    feet = (1 << m.names.index("left_ankle_roll_link")) | (1 << m.names.index("right_ankle_roll_link"))
    bits = ent.contact_bits.cpu().numpy()
    flag = ent.contact.cpu().numpy()
    assert np.all(bits[:4] == feet) and np.all(flag[:4] == 0)
    assert np.abs(s.root_pos[:4, 2] - z0).max() < 0.01 and np.abs(s.root_pos[:4, :2]).max() < 0.03 and s.root_quat[:4, 0].min() > 0.9998
    assert np.all(flag[4:] == 1) and s.root_pos[4:, 2].max() < 0.3 and np.abs(s.root_vel[4:]).max() < 0.5
    assert np.isfinite(ent.pose.cpu().numpy()).all() and np.isfinite(ent.vel.cpu().numpy()).all()
    lp = ent.get_links_pos().cpu().numpy()
    assert lp.shape == (n, 30, 3) and np.abs(lp[:4, m.names.index("left_ankle_roll_link"), 2] - 0.035).max() < 0.02


def test_training_iteration_on_the_rigid_engine():
    """engine=rigid behind the unchanged agent: rollout + update run, falls terminate episodes through the engine's own
    contact predicate, everything stays finite."""
    import torch
    import add_gym_amd.learning.add_agent as A
    from add_gym_amd.config import load_config

    cfg = load_config("train", ["engine=rigid", "engine.num_envs=512", "agent.steps_per_iter=16", "task.motion_file=synthetic:2x120"])
    cfg["task"]["motion_joint_order"] = kin_meta()["motion_joint_order"]
    ag = A.ADDAgent(cfg)
    assert ag._fast_engine and type(ag._env.robot.entity).__name__ == "RigidEntity"
    ag.reset_all_envs()
    ag._init_train()
    for _ in range(3):
        info = ag._train_iter()
        ag._iter += 1
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in info.values())
    done = ag._B["done"].cpu().numpy()
    assert (done == 1).sum() > 0          # an untrained policy falls: FAIL flags come from the engine's contact / pose tests
    assert torch.isfinite(ag._S["sim_pose"]).all() and torch.isfinite(ag._S["sim_vel"]).all()
    z = ag._S["sim_pose"][:, 2]
    assert float(z.min()) > -0.05 and float(z.max()) < 1.5  # nobody sank through the floor or flew away


def test_domain_randomisation_per_env_gains_friction_and_pushes():
    """engine.domain_randomization (build-defined extension; the reference has none): per-env PD gain scale and friction reach the
    kernel (parity with the oracle given the same per-env values), the ranges are honoured, pushes kick the root."""
    import torch

    n = 64
    dr = dict(enabled=True, seed=3, gain_scale=[0.7, 1.3], friction=[0.4, 1.2], resample_interval=0, push_interval=5, push_velocity=0.6)
    eng, scene, plane, ent, m, kp, kv = make_entity(n, domain_randomization=dr)
    sc = ent.env_scale.cpu().numpy().astype(np.float64)
    assert sc.shape == (n, 2) and 0.7 <= sc[:, 0].min() < sc[:, 0].max() <= 1.3 and 0.4 <= sc[:, 1].min() < sc[:, 1].max() <= 1.2
    rng = np.random.RandomState(9)
    st = rand_states(rng, n, 0.25, 0.8)
    pose, vel = (a.astype(F).astype(np.float64) for a in st.packed())
    st = RB.State.from_packed(pose, vel)
    tgt = rng.uniform(-0.5, 0.5, (n, 29)).astype(F)
    put(ent, st)
    ent.control_dofs_position(torch.tensor(tgt, device="cuda"))
    scene.step()
    torch.cuda.synchronize()
    got = get(ent)
    want, touch = RB.step(m, RB.RigidParams(friction=sc[:, 1]), kp * sc[:, :1], kv * sc[:, :1], st, tgt.astype(np.float64))
    assert touch.sum() > 100
    for name in ("root_pos", "q", "root_vel", "root_ang", "qd"):
        a, b = getattr(got, name), getattr(want, name)
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max()), name
    # the same states WITHOUT randomisation evolve differently
    eng2, scene2, plane2, ent2, *_ = make_entity(n)
    put(ent2, st)
    ent2.control_dofs_position(torch.tensor(tgt, device="cuda"))
    scene2.step()
    torch.cuda.synchronize()
    assert np.abs(get(ent2).qd - got.qd).max() > 1e-2
    # the draws themselves: addhip_rigid_randomize at control-step index s redraws env_scale from Philox stream (8<<40)+s and kicks the
    # root from stream (9<<40)+s -- the numbers addhip_fill_uniform gives for those streams -- when s is due, and advances the
    # device-resident counter either way (odd env count: the last Philox call serves one env only)
    import add_gym_amd._lib as L

    n2 = 77
    drc = L.RigidDrT(3, 4, 5, 0.7, 1.3, 0.4, 1.2, 0.6)  # seed, resample every 4, push every 5
    scale = torch.full((n2, 2), -1.0, device="cuda")
    vel = torch.randn(n2, 36, device="cuda")
    ctr = torch.zeros(2, dtype=torch.int64, device="cuda")
    u = torch.zeros(n2, 2, device="cuda")
    lo, width = torch.tensor([0.7, 0.4], device="cuda"), torch.tensor([0.6, 0.8], device="cuda")
    st_ = L.current_stream()
    L.call("addhip_rigid_randomize", drc, L.ptr(scale), L.ptr(vel), n2, L.ptr(ctr), 0, st_)  # initialisation: draws, does not advance
    L.call("addhip_fill_uniform", L.ptr(u), 2 * n2, 3, (8 << 40) + 0, st_)
    torch.testing.assert_close(scale, lo + u * width, rtol=0, atol=1e-6)
    assert ctr.tolist() == [0, 0]
    for s_ in range(0, 11):
        scale0, vel0 = scale.clone(), vel.clone()
        L.call("addhip_rigid_randomize", drc, L.ptr(scale), L.ptr(vel), n2, L.ptr(ctr), 1, st_)
        assert ctr.tolist() == [s_ + 1, 0]
        if s_ > 0 and s_ % 4 == 0:
            L.call("addhip_fill_uniform", L.ptr(u), 2 * n2, 3, (8 << 40) + s_, st_)
            torch.testing.assert_close(scale, lo + u * width, rtol=0, atol=1e-6)
            assert not torch.equal(scale, scale0)
        else:
            assert torch.equal(scale, scale0)
        if s_ > 0 and s_ % 5 == 0:
            L.call("addhip_fill_uniform", L.ptr(u), 2 * n2, 3, (9 << 40) + s_, st_)
            torch.testing.assert_close(vel[:, 0:2], vel0[:, 0:2] + (2.0 * u - 1.0) * 0.6, rtol=0, atol=1e-6)
            assert float((vel[:, 0:2] - vel0[:, 0:2]).abs().max()) > 0.5
        else:
            assert torch.equal(vel[:, 0:2], vel0[:, 0:2])
        assert torch.equal(vel[:, 2:], vel0[:, 2:])
    # and through the engine: the entity's own counter follows its control steps; step index 5 kicks the root before the physics step
    for _ in range(4):
        scene.step()
    torch.cuda.synchronize()
    assert ent._d_steps.tolist() == [5, 0]
    v_before = ent.vel[:, 0:2].clone()
    scene.step()
    torch.cuda.synchronize()
    dv = (ent.vel[:, 0:2] - v_before).abs().max(dim=1).values
    assert float(dv.max()) > 0.3 and torch.isfinite(ent.vel).all() and ent._d_steps.tolist() == [6, 0]


def test_gain_change_refreshes_the_device_tables_in_place():
    """set_dofs_kp / set_termination_links after build: the next step uploads into the SAME device buffers (addresses baked into a
    captured rollout stay valid) and bumps model_version (holders of captured launches re-capture: the scalar fields travel by value)."""
    import torch

    eng, scene, plane, ent, m, kp, kv = make_entity(8)
    scene.step()
    ptrs, ver = (ent._d_body.data_ptr(), ent._d_topo.data_ptr(), ent._d_points.data_ptr(), ent._d_chains.data_ptr()), ent.model_version
    body0 = ent._d_body.clone()
    ent.set_dofs_kp(torch.cat([torch.zeros(6), torch.tensor(kp * 2.0, dtype=torch.float32)]))
    ent.set_termination_links(["left_ankle_roll_link", "right_ankle_roll_link"])
    assert ent._dirty
    scene.step()
    torch.cuda.synchronize()
    assert ptrs == (ent._d_body.data_ptr(), ent._d_topo.data_ptr(), ent._d_points.data_ptr(), ent._d_chains.data_ptr())
    assert ent.model_version == ver + 1 and not ent._dirty
    assert torch.equal(ent._d_body[:, 27], body0[:, 27] * 2.0) and torch.equal(ent._d_body[:, :27], body0[:, :27])

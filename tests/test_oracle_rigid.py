"""The float64 oracle of the rigid-body step (oracle/rigid.py) against physics it cannot fake: the articulated-body
result must satisfy inverse dynamics (recursive Newton-Euler, an independent algorithm), and a free-floating robot must
conserve momentum and energy and fall like its centre of mass (errors vanish to first order with the step).  CPU only.
These pin the ALGORITHM (the reference holds no physics fixtures: parity unpinned against Genesis / MuJoCo)."""
import copy

import numpy as np

from oracle import rigid as RB
from tests.util import G1_XML


def model():
    return RB.RigidModel(G1_XML)


def rand_state(rng, n, z):
    q = rng.uniform(-0.3, 0.3, (n, 29))
    quat = rng.standard_normal((n, 4)) * 0.2 + np.array([1.0, 0, 0, 0])
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    return RB.State(np.c_[rng.uniform(-1, 1, (n, 2)), np.full(n, z)], quat, q, rng.standard_normal((n, 3)) * 0.5, rng.standard_normal((n, 3)),
                    rng.standard_normal((n, 29)) * 2)


def test_model_tables():
    m = model()
    assert m.nb == 30 and abs(m.total_mass - 35.11) < 0.05  # G1 29-dof: 35 kg
    assert m.names[0] == "pelvis" and m.parent[0] == -1 and all(m.parent[i] < i for i in range(1, m.nb))
    assert len(m.pt_body) > 100 and set(m.pt_body[m.pt_rad > 0.004]) >= {m.names.index("left_ankle_roll_link"), m.names.index("right_ankle_roll_link")}
    kp, kv = RB.gains(m)
    assert kp[m.names.index("left_knee_link") - 1] == 144.0 and kp[m.names.index("left_hip_yaw_link") - 1] == 96.0
    assert kp[m.names.index("torso_link") - 1] == 60.0 and np.allclose(kv, 2 * np.sqrt(kp))


def test_articulated_body_result_satisfies_inverse_dynamics_with_contacts():
    m, prm = model(), RB.RigidParams()
    kp, kv = RB.gains(m)
    rng = np.random.RandomState(0)
    st = rand_state(rng, 16, 0.5)  # low enough for several links to be in contact
    tgt = rng.uniform(-0.3, 0.3, (16, 29))
    a0, qdd, info = RB.accelerations(m, prm, kp, kv, st, tgt, 0.0025)
    assert info["touching"].sum() > 20
    # what acts on the bodies: the linearly-implicit contact wrench f0 - A a_i; what acts on the joints: tau - dadd qdd
    Fext = info["Fc"] - np.einsum("nbij,nbj->nbi", info["Ac"], info["a"])
    tau, root_wrench = RB.inverse_dynamics(m, st, a0, qdd, Fext)
    want = info["tau"] - info["dadd"] * qdd
    assert np.abs(tau - want).max() <= 1e-9 * np.abs(want).max()
    assert np.abs(root_wrench).max() <= 1e-9 * np.abs(want).max()   # nothing pushes the floating base


def test_free_flight_conserves_momentum_and_energy_to_first_order():
    m = copy.deepcopy(model())
    m.damping = m.damping * 0.0
    rng = np.random.RandomState(1)
    st0 = rand_state(rng, 4, 3.0)
    z = np.zeros(29)
    P0, L0, ke0, pe0, com0 = RB.momentum_and_energy(m, st0)
    E0 = ke0 + pe0 + 0.5 * (m.armature * st0.qd ** 2).sum(1)  # rotor inertia (armature) carries kinetic energy too
    Lc0 = L0 - np.cross(com0, P0 )
    T = 0.1
    errs = []
    for h in (0.001, 0.0005):
        prm = RB.RigidParams(dt=h, substeps=1, limit_stiffness=0.0)
        s = st0
        for _ in range(int(round(T / h))):
            s, _ = RB.step(m, prm, z, z, s, np.zeros((4, 29)), contacts=False)
        P, L, ke, pe, com = RB.momentum_and_energy(m, s)
        E = ke + pe + 0.5 * (m.armature * s.qd ** 2).sum(1)
        g = np.array([0, 0, -m.total_mass * RB.GRAVITY * T])
        errs.append((np.abs(P - (P0 + g)).max(), np.abs((L - np.cross(com, P)) - Lc0).max(), np.abs(E - E0).max(),
                     np.abs(com[:, 2] - (com0[:, 2] + P0[:, 2] / m.total_mass * T - 0.5 * RB.GRAVITY * T * T)).max()))
    (p1, l1, e1, c1), (p2, l2, e2, c2) = errs
    scaleP, scaleL, scaleE = m.total_mass * RB.GRAVITY * T, np.abs(Lc0).max(), ke0.max()
    assert p2 < 2e-3 * scaleP and l2 < 4e-3 * scaleL and e2 < 4e-3 * scaleE and c2 < 5e-4
    for a, b in ((p1, p2), (l1, l2), (e1, e2)):   # halving the step halves the error: first-order integrator, nothing systematic
        assert 1.6 < a / b < 2.4, (a, b)


def test_stand_holds_on_stiff_ankles_and_tips_slowly_on_the_robot_gains():
    m = copy.deepcopy(model())
    st = RB.State(np.array([[0, 0, 0.8]]), np.array([[1.0, 0, 0, 0]]), np.zeros((1, 29)), np.zeros((1, 3)), np.zeros((1, 3)), np.zeros((1, 29)))
    R, p, _ = RB.forward_kinematics(m, st.root_pos, st.root_quat, st.q)
    zmin = min(p[0, b, 2] + R[0, b, 2, :] @ r - rad for b, r, rad in zip(m.pt_body, m.pt_pos, m.pt_rad))
    st.root_pos[:, 2] -= zmin - 0.0005
    feet = {m.names.index("left_ankle_roll_link"), m.names.index("right_ankle_roll_link")}
    # (i) the robot's own gains (robot.py:133-163): ankle stiffness 144 N m/rad < m g l ~ 240: an inverted pendulum, tips slowly
    kp, kv = RB.gains(m)
    s = st.copy()
    for _ in range(30):
        s, touch = RB.step(m, RB.RigidParams(), kp, kv, s, np.zeros((1, 29)))
    assert set(np.nonzero(touch[0])[0]) == feet and abs(s.root_pos[0, 2] - st.root_pos[0, 2]) < 0.01 and abs(s.root_quat[0, 0]) > 0.9995
    # (ii) 5x stiffer joints, no torque clamp: stands for 2 s, feet on the ground, no sinking, no sliding
    m.frc_limit[:] = 1e9
    kp, kv = RB.gains(m, 6.0)
    s = st.copy()
    for _ in range(200):
        s, touch = RB.step(m, RB.RigidParams(max_torque=1e9), kp, kv, s, np.zeros((1, 29)))
    assert set(np.nonzero(touch[0])[0]) == feet
    assert abs(s.root_pos[0, 2] - st.root_pos[0, 2]) < 0.01 and np.abs(s.root_pos[0, :2]).max() < 0.03 and s.root_quat[0, 0] > 0.9998
    assert np.abs(s.q).max() < 0.05 and np.abs(s.root_vel).max() < 0.05
    # the same trajectory with one physics step per control step (h = 10 ms) instead of four: the implicit terms keep it stable
    s1 = st.copy()
    for _ in range(200):
        s1, _ = RB.step(m, RB.RigidParams(max_torque=1e9, substeps=1), kp, kv, s1, np.zeros((1, 29)))
    assert abs(s1.root_pos[0, 2] - s.root_pos[0, 2]) < 0.005 and s1.root_quat[0, 0] > 0.9998


def test_chain_table_of_the_four_lane_kernel():
    """RigidModelTables.chain_table(): G1 cuts into left leg | right leg | waist + left arm | right arm (the right arm hangs off the
    torso = step 2 of lane 2, so it starts at step 3); every body appears once, in depth-first order along its chain; a tree
    that needs more than 10 steps or 4 chains has no table (the one-lane kernel handles it)."""
    import add_gym_amd  # noqa: F401
    from add_gym_amd.engine.rigid_model import RigidModelTables

    t = RigidModelTables(G1_XML)
    tab = t.chain_table()
    assert tab.shape == (4, 16) and tab.dtype == np.int32
    assert tab[:, 0].tolist() == [6, 6, 10, 7] and tab[:, 1].tolist() == [0, 0, 0, 3] and tab[:, 2].tolist() == [-1, -1, -1, 2]
    bodies = np.concatenate([tab[q, 3:3 + tab[q, 0]] for q in range(4)])
    assert sorted(bodies.tolist()) == list(range(1, t.num_bodies))
    for q in range(4):
        run = tab[q, 3:3 + tab[q, 0]]
        assert np.all(np.diff(run) == 1) and all(t.topo[k, 0] == k - 1 for k in run[1:])
        par = int(t.topo[run[0], 0])
        if tab[q, 2] < 0:
            assert par == 0
        else:
            a = int(tab[q, 2])
            assert par == tab[a, 3 + (tab[q, 1] - 1 - tab[a, 1])]  # the attach body sits one step before this chain's start
    assert (tab[:, 0] + tab[:, 1]).max() <= RigidModelTables.MAX_CHAIN_STEPS
    t2 = copy.copy(t)
    t2.MAX_CHAIN_STEPS = 9
    assert t2.chain_table() is None

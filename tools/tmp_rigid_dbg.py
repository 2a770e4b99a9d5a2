import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_hip_rigid import make_entity, rand_states, put, get
from oracle import rigid as RB
n = 16
res = {}
for lanes in (1, 4):
    eng, scene, plane, ent, m, kp, kv = make_entity(n, lanes_per_env=lanes, substeps=1)
    rng = np.random.RandomState(4)
    st = rand_states(rng, n, 2.0, 3.0)
    pose, vel = (a.astype(np.float32).astype(np.float64) for a in st.packed())
    mode = os.environ.get("MODE", "rand")
    if mode == "zero":
        pose[:, 7:] = 0; vel[:] = 0; pose[:, 3:7] = [1, 0, 0, 0]
    if mode == "q":
        vel[:] = 0; pose[:, 3:7] = [1, 0, 0, 0]
    if mode == "q1":
        vel[:] = 0; pose[:, 3:7] = [1, 0, 0, 0]; keep = int(os.environ.get("DOF", "0")); x = pose[:, 7 + keep].copy(); pose[:, 7:] = 0; pose[:, 7 + keep] = x
    if mode == "v1":
        pose[:, 7:] = 0; pose[:, 3:7] = [1, 0, 0, 0]; keep = int(os.environ.get("DOF", "0")); x = vel[:, 6 + keep].copy(); vel[:] = 0; vel[:, 6 + keep] = x
    if mode == "v":
        pose[:, 7:] = 0; pose[:, 3:7] = [1, 0, 0, 0]; vel[:, :6] = 0
    st = RB.State.from_packed(pose, vel)
    tgt = rng.uniform(-0.5, 0.5, (n, 29)).astype(np.float32) if mode == "rand" else pose[:, 7:36].astype(np.float32)
    put(ent, st)
    ent.control_dofs_position(torch.tensor(tgt, device="cuda"))
    scene.step(); torch.cuda.synchronize()
    res[lanes] = (ent.pose.cpu().numpy().copy(), ent.vel.cpu().numpy().copy())
dp = np.abs(res[1][0] - res[4][0]).max(0); dv = np.abs(res[1][1] - res[4][1]).max(0)
np.set_printoptions(precision=2, linewidth=200)
print("dofs with vel diff > 1e-4:", np.nonzero(dv[6:35] > 1e-4)[0], "root", dv[:6])


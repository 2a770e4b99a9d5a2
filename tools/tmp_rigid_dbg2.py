import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_hip_rigid import make_entity, rand_states, put, get
from oracle import rigid as RB
n = 16
ents = {l: make_entity(n, lanes_per_env=l, substeps=1) for l in (1, 4)}
rng = np.random.RandomState(4)
st0 = rand_states(rng, n, 2.0, 3.0)
pose0, vel0 = (a.astype(np.float32).astype(np.float64) for a in st0.packed())
print('max |qd|', np.abs(vel0[:, 6:35]).max())
for d in (0, 8, 12):
    out = {}
    for l in (1, 4):
        eng, scene, plane, ent, m, kp, kv = ents[l]
        pose, vel = pose0.copy(), vel0.copy()
        pose[:, 7:] = 0; pose[:, 3:7] = [1, 0, 0, 0]; x = vel[:, 6 + d].copy() * float(os.environ.get('VS', '1')); vel[:] = 0; vel[:, 6 + d] = x
        put(ent, RB.State.from_packed(pose, vel))
        ent.control_dofs_position(torch.tensor(pose[:, 7:36].astype(np.float32), device="cuda"))
        scene.step(); torch.cuda.synchronize()
        out[l] = ent.vel.cpu().numpy().copy()
    dv = np.abs(out[1] - out[4]).max(0)
    eng, scene, plane, ent, m, kp, kv = ents[1]
    want, touch = RB.step(m, RB.RigidParams(substeps=1), kp, kv, RB.State.from_packed(pose, vel), pose[:, 7:36])
    wv = want.packed()[1]
    print(d, "max vel diff %.2e" % dv.max(), "at col", int(dv.argmax()), "| 1-lane vs oracle %.2e, 4-lane vs oracle %.2e" % (np.abs(out[1] - wv).max(), np.abs(out[4] - wv).max()))

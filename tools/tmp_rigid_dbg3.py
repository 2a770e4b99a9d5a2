import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_hip_rigid import make_entity, rand_states, put, get
from oracle import rigid as RB
import add_gym_amd._lib as L
n = 16
perm = [int(c) for c in os.environ.get("PERM", "0123")]
ents = {l: make_entity(n, lanes_per_env=l, substeps=1) for l in (1, 4)}
e4 = ents[4][3]
ents[4][1].step(); torch.cuda.synchronize()  # (the pending gain upload rebuilds the tables: do it before overriding)
tab = e4.tables.chain_table()
new = np.zeros_like(tab); new[:, 2] = -1
for newlane, old in enumerate(perm):
    new[newlane] = tab[old]
    if new[newlane, 2] >= 0:
        new[newlane, 2] = perm.index(int(tab[old, 2]))
print(new[:, :4])
e4._d_chains = torch.tensor(new, device="cuda"); e4.c_struct.chains = L.ptr(e4._d_chains)
rng = np.random.RandomState(4)
st0 = rand_states(rng, n, 2.0, 3.0)
pose0, vel0 = (a.astype(np.float32).astype(np.float64) for a in st0.packed())
for d in (0, 1, 8, 12, 22, 9):
    out = {}
    for l in (1, 4):
        eng, scene, plane, ent, m, kp, kv = ents[l]
        pose, vel = pose0.copy(), vel0.copy()
        pose[:, 7:] = 0; pose[:, 3:7] = [1, 0, 0, 0]; x = vel[:, 6 + d].copy(); vel[:] = 0; vel[:, 6 + d] = x
        put(ent, RB.State.from_packed(pose, vel))
        ent.control_dofs_position(torch.tensor(pose[:, 7:36].astype(np.float32), device="cuda"))
        scene.step(); torch.cuda.synchronize()
        out[l] = ent.vel.cpu().numpy().copy()
    dv = np.abs(out[1] - out[4]).max(0)
    print(d, "max vel diff %.2e" % dv.max(), "cols>1e-4:", np.nonzero(dv > 1e-4)[0])

import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_hip_rigid import make_entity, rand_states, put, get
from oracle import rigid as RB
import add_gym_amd._lib as L
n = 16
d = int(os.environ.get("DOF", "0"))
dump = {}
for lanes in (1, 4):
    eng, scene, plane, ent, m, kp, kv = make_entity(n, lanes_per_env=lanes, substeps=1)
    rng = np.random.RandomState(4)
    st0 = rand_states(rng, n, 2.0, 3.0)
    pose, vel = (a.astype(np.float32).astype(np.float64) for a in st0.packed())
    pose[:, 7:] = 0; pose[:, 3:7] = [1, 0, 0, 0]; x = vel[:, 6 + d].copy(); vel[:] = 0; vel[:, 6 + d] = x
    put(ent, RB.State.from_packed(pose, vel))
    ent.control_dofs_position(torch.tensor(pose[:, 7:36].astype(np.float32), device="cuda"))
    scene.step(); torch.cuda.synchronize()
    buf = (C.c_float * 1024)()
    lib = L.load(); lib.addhip_dbg_read.argtypes = [C.c_void_p]
    lib.addhip_dbg_read(buf)
    dump[lanes] = np.array(buf).reshape(32, 32).copy()
t = ent.tables
print('lane info (rw.y, sub, substeps, ww.y, clen, cstart):', dump[4][0][:24].reshape(4, 6))
np.set_printoptions(precision=5, linewidth=220, suppress=True)
names = ["w", "w", "w", "vl", "vl", "vl", "p0a", "p0a", "p0a", "p0l", "p0l", "p0l", "u", "Dinv", "ppa", "ppa", "ppa", "ppl", "ppl", "ppl", "tau", "PAxx", "PCxx", "pa", "pa", "pa", "qdd", "aa", "aa", "aa", "alx", "alz"]
for k in range(1, 30):
    a, b = dump[1][k], dump[4][k]
    diff = np.abs(a - b)
    bad = np.nonzero(diff > 1e-5 * (1 + np.abs(a)))[0]
    print(k, t.names[t.bfs_of_traversal[k]][:24].ljust(24), "differs in:", [(names[j], float(a[j]), float(b[j])) for j in bad][:6])

import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_hip_rigid import make_entity, rand_states, put, get
from oracle import rigid as RB
import add_gym_amd._lib as L
n = 16
eng, scene, plane, ent, m, kp, kv = make_entity(n, lanes_per_env=4, substeps=1)
rng = np.random.RandomState(4)
st0 = rand_states(rng, n, 2.0, 3.0)
pose, vel = (a.astype(np.float32).astype(np.float64) for a in st0.packed())
d = int(os.environ.get("DOF", "0"))
pose[:, 7:] = 0; pose[:, 3:7] = [1, 0, 0, 0]; x = vel[:, 6 + d].copy(); vel[:] = 0; vel[:, 6 + d] = x
put(ent, RB.State.from_packed(pose, vel))
ent.control_dofs_position(torch.tensor(pose[:, 7:36].astype(np.float32), device="cuda"))
scene.step(); torch.cuda.synchronize()
buf = (C.c_float * 640)()
lib = L.load(); lib.addhip_dbg_read.argtypes = [C.c_void_p]
print("rc", lib.addhip_dbg_read(buf))
a = np.array(buf).reshape(4, 10, 16)
np.set_printoptions(precision=4, linewidth=220, suppress=True)
t = ent.tables
print("qd of env 0:", x[0])
for q in range(4):
    for i in range(10):
        r = a[q, i]
        if r[12] == 0 and i > 0: continue
        k = int(r[12]); bc = t.body[k]
        w, vl = r[0:3].astype(np.float64), r[3:6].astype(np.float64)
        A = np.array([[bc[16], bc[17], bc[18]], [bc[17], bc[19], bc[20]], [bc[18], bc[20], bc[21]]], np.float64)
        mc = bc[13:16].astype(np.float64); mass = float(bc[12])
        hn = A @ w + np.cross(mc, vl); hl = mass * vl - np.cross(mc, w)
        nzv = np.array([0, 0, 1.0]); gl = -9.81 * nzv
        pa = np.cross(w, hn) + np.cross(vl, hl) - np.cross(mc, gl); pl_ = np.cross(w, hl) - mass * gl
        print(q, i, "k", k, t.names[t.bfs_of_traversal[k]][:22].ljust(22), "ax", int(r[14]), "qd", r[13], "w", r[0:3], "vl", r[3:6], "| p.a err", np.abs(pa - r[6:9]).max(), "p.l err", np.abs(pl_ - r[9:12]).max())

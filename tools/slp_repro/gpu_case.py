"""One variant of libaddhip.so against the float64 oracle on the 70-state case of tests/test_hip_rigid.py (four lanes per env, LDS form;
`regs` = the register form on the tiled case).  Prints one line per case: worst |error| / scale per quantity.  Used by gpu_bisect.py.

    python tools/slp_repro/gpu_case.py <path to a libaddhip.so variant> [regs]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import add_gym_amd  # noqa: E402,F401
from add_gym_amd import _lib as L  # noqa: E402

L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch  # noqa: E402

from oracle import rigid as RB  # noqa: E402
from tests.test_hip_rigid import make_entity, rand_states  # noqa: E402

F = np.float32
regs = len(sys.argv) > 2 and sys.argv[2] == "regs"
for case in ("contact", "flight"):
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    reps = (2 * 16 * cus) // 70 + 2 if regs else 1
    n = 70 * reps - (3 if regs else 0)
    eng, scene, plane, ent, m, kp, kv = make_entity(n, lanes_per_env=4)
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
    got = RB.State.from_packed(ent.pose.cpu().numpy().astype(np.float64), ent.vel.cpu().numpy().astype(np.float64))
    want, touch = RB.step(m, RB.RigidParams(), kp, kv, RB.State.from_packed(pose.astype(np.float64), vel.astype(np.float64)), tgt.astype(np.float64))
    errs = {}
    for name in ("root_pos", "root_quat", "q", "root_vel", "root_ang", "qd"):
        a, b = getattr(got, name)[:70], getattr(want, name)
        errs[name] = float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    tol = 1e-5 * (10 if case == "contact" else 1)
    worst = max(errs.values())
    print(f"CASE {case} {'regs' if regs else 'lds'} worst {worst:.3e} tol {tol:.0e} {'ok' if worst <= tol else 'WRONG'} " +
          " ".join(f"{k}={v:.1e}" for k, v in errs.items()), flush=True)

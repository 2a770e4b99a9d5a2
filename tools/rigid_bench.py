"""Rigid-body engine alone: time of one control step (addhip_rigid_step: `substeps` articulated-body sweeps) per env count, HIP
events on the launch stream; states are kept plausible (standing robots, PD targets = small random offsets, reset every chunk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import add_gym_amd
import add_gym_amd._lib as L
from tests.test_hip_rigid import make_entity

lanes = int(os.environ.get("LANES", "4"))  # 4: one lane per chain of the tree (default), 1: the one-lane kernel
print(f"lanes per env: {lanes}")
for n in [int(a) for a in sys.argv[1:]] or [4096, 16384, 65536]:
    eng, scene, plane, ent, m, kp, kv = make_entity(n, lanes_per_env=lanes)
    pose0 = ent.pose.clone(); pose0[:, 2] = 0.79
    tgt = (torch.randn(n, 32, device="cuda") * 0.1).contiguous()
    ent.control_dofs_position(tgt)
    st = torch.cuda.current_stream()
    tot, cnt = 0.0, 0
    for chunk in range(6):
        ent.pose.copy_(pose0); ent.vel.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            scene.step()
        e1.record(st); e1.synchronize()
        if chunk:
            tot += e0.elapsed_time(e1); cnt += 20
    ms = tot / cnt
    sub = int(ent._opts["substeps"])
    print(f"envs {n:6d}: {ms*1e3:8.1f} us per control step ({sub} substeps) = {n/ms/1e3:8.2f} M env-steps/s; {ms*1e3/sub:7.1f} us per substep; "
          f"in contact: {float((ent.contact_bits != 0).float().mean()):.2f}", flush=True)

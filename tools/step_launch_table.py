"""Every launch of one optimiser step (the recorded update plan at 4096 envs -> 16 384-row minibatch) replayed alone: entry point, HIP-event
time (median of 7).  The GEMM launches are tools/gemm_step_table.py's; this lists the loss heads, reductions and the other helpers beside them.

    python tools/step_launch_table.py [precision] [config overrides ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import add_gym_amd  # noqa: F401
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ag = ADDAgent(load_config("train", ["engine.num_envs=4096", f"agent.matmul_precision={prec}"] + sys.argv[2:]))
for w in ag._W.values():
    if w.dtype == torch.float32:
        w.normal_()
ag._W["mb_mask"].fill_(1.0)
for r in (ag._run_actor, ag._run_critic, ag._run_disc):
    for t in r.h + r.dz:
        t.normal_()
st = torch.cuda.current_stream()
_w = torch.randn(8192, 8192, device="cuda")
for _ in range(60):
    _w @ _w
torch.cuda.synchronize()
plan = ag._update_plan
tot = {}
for i, (name, gemms) in enumerate(plan.launches()):
    ts = []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        plan.run(st.cuda_stream, i, i + 1)
        e1.record(st)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    us = sorted(ts[1:])[3] * 1e3
    shape = " ".join(f"{g.M}x{g.N}x{g.K}" + (f"/{g.split_k}" if g.split_k > 1 else "") for g in gemms)
    tot[name] = tot.get(name, 0.0) + us
    print(f"{i:3d} {name:28s} {us:8.1f} us  {shape}", flush=True)
print("by entry point:", ", ".join(f"{k} {v:.0f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])), f"| total {sum(tot.values()):.0f} us")

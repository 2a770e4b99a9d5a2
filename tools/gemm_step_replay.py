"""Replays only the GEMM launches of one optimiser step (N=4096 envs -> 16384-row minibatch) three times, for PMC passes.
usage: gemm_step_replay.py [fp32|bf16x3|f16x2|bf16x2|bf16] [envs (default 4096)]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ag = ADDAgent(load_config("train", [f"engine.num_envs={envs}", f"agent.matmul_precision={prec}", "task.motion_file=synthetic:1x300"]))
for w in ag._W.values():
    if w.dtype == torch.float32: w.normal_()
for r in (ag._run_actor, ag._run_critic, ag._run_disc):
    for t in r.h + r.dz: t.normal_()
    if getattr(r, "storage16", 0):
        for t in r.h16 + r.dz16: t.copy_(torch.randn(t.shape, device=t.device))
for k, w in ag._W.items():
    if w.dtype == torch.bfloat16: w.copy_(torch.randn(w.shape, device=w.device))
st = torch.cuda.current_stream().cuda_stream
plan = ag._update_plan
calls = [i for i, (name, gemms) in enumerate(plan.launches()) if gemms or name == "addhip_actor_head"]  # (the head section: three 32-wide products in one launch)
print("gemm launches per step:", len(calls))
for _ in range(3):
    for i in calls:
        plan.run(st, i, i + 1)
torch.cuda.synchronize()

"""Experiment: one optimiser step (gather + three-stream sections + AdamW + bf16 shadow refresh) call by call vs replayed from a
hipGraph, and the host-side enqueue time of the call-by-call path.  Timing only (the captured AdamW bakes its step count in)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ag = ADDAgent(load_config("train", [f"engine.num_envs={N}", f"agent.matmul_precision={prec}", "task.motion_file=synthetic:1x3600"]))
ag.reset_all_envs(); ag._init_train(); ag._train_iter()
m = ag._model
def step_plain():
    st = L.current_stream()
    L.call("addhip_gather_minibatch", ag._gather_c, st)
    ag._run_update_sections()
    L.call("addhip_adamw", L.ptr(m.params), L.ptr(m.grads), L.ptr(m.exp_avg), L.ptr(m.exp_avg_sq), m.count, 0.0, 0.9, 0.999, 1e-8, 0.0, 100, st)
    if ag._storage16:
        m.refresh_shadow(st)
for _ in range(3): step_plain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): step_plain()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{prec} N={N} plain: {(t2 - t0) * 25:.3f} ms per step; host enqueue alone {(t1 - t0) * 25:.3f} ms per step", flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step_plain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        step_plain()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): g.replay()
torch.cuda.synchronize()
print(f"{prec} N={N} graph: {(time.perf_counter() - t0) * 25:.3f} ms per step", flush=True)

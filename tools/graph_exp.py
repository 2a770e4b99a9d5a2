"""Experiment: the three-stream update sections of one optimiser step captured into a hipGraph vs launched call by call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ag = ADDAgent(load_config("train", ["engine.num_envs=4096", f"agent.matmul_precision={prec}", "task.motion_file=synthetic:1x3600"]))
ag.reset_all_envs(); ag._init_train(); ag._train_iter()
def step_plain():
    L.call("addhip_gather_minibatch", ag._gather_c, L.current_stream())
    ag._run_update_sections()
for _ in range(3): step_plain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40): step_plain()
torch.cuda.synchronize()
print(f"{prec} plain: {(time.perf_counter() - t0) * 25:.3f} ms per step", flush=True)
ref = ag._model.grads.clone()
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
print(f"{prec} graph: {(time.perf_counter() - t0) * 25:.3f} ms per step", flush=True)
d = (ag._model.grads - ref).abs().max().item() / ref.abs().max().item()
print("max relative gradient difference graph vs plain (atomics reorder only):", d)

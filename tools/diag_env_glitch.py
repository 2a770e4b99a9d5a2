"""Diagnostic: per-chunk wall / event / per-call host time of addhip_env_step (looks for intermittent ms-scale stalls)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = load_config("train", [f"engine.num_envs={N}", "agent.steps_per_iter=2", "agent.batch_size=1"])
ag = ADDAgent(cfg); ag.reset_all_envs(); ag._init_train()
st = torch.cuda.current_stream()
out = ag._step_out[0]
for chunk in range(8):
    if chunk % 2 == 0:
        ag.reset_all_envs()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    host = []
    t0 = time.perf_counter()
    evs[0].record(st)
    for i in range(10):
        h0 = time.perf_counter()
        L.call("addhip_env_step", ag._motion_lib.c_struct, ag._task, ag._env_c, out, 0, st.cuda_stream)
        host.append((time.perf_counter() - h0) * 1e6)
        evs[i + 1].record(st)
    evs[-1].synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    per = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(10)]
    done = int((ag._S["done"] != 0).sum())
    print(f"chunk {chunk}: wall {wall:9.1f} us  events {sum(per):9.1f} us  per-step {[round(x) for x in per]}  host-call max {max(host):7.1f} us  done envs {done}", flush=True)

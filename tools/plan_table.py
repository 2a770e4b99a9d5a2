"""Every call of the update plan of one optimiser step, timed in isolation (HIP events, median of 7): name, time.  Sums by kernel."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ag = ADDAgent(load_config("train", ["engine.num_envs=4096", f"agent.matmul_precision={prec}"]))
ag.reset_all_envs(); ag._init_train(); ag._train_iter()   # realistic buffer contents
st = torch.cuda.current_stream()
tot = collections.defaultdict(lambda: [0, 0.0])
plan = ag._update_plan
for i, (name, gemms) in enumerate(plan.launches()):
    ts = []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); plan.run(st.cuda_stream, i, i + 1); e1.record(st); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts[1:])[3]
    tot[name][0] += 1; tot[name][1] += us
    if "-v" in sys.argv: print(f"{i:3d} {name:28s} {us:8.1f} us")
print(f"precision {prec}: launches per optimiser step = {len(plan)}")
for name, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:28s} x{n:3d} {us:9.1f} us")
print(f"{'total':28s} {sum(v[1] for v in tot.values()):14.1f} us")

"""Where the 1-rank exchange path spends its extra time: update phase wall time and host enqueue time, plain vs a 1-rank RCCL group
(every optimiser step then issues four asynchronous bucket all-reduces from the schedule's call-backs).  usage: exchange_probe.py [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
def run(distributed):
    ag = ADDAgent(load_config("train", ["engine.num_envs=4096", f"agent.matmul_precision={prec}", "task.motion_file=synthetic:1x3600"]), distributed=distributed)
    ag.reset_all_envs(); ag._init_train()
    for _ in range(2): ag._train_iter(); ag._iter += 1
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        ag._rollout_train(); ag._build_train_data(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ag._update_model(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        res.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
    return min(r[0] for r in res), min(r[1] for r in res)
h0, w0 = run(False)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl")
h1, w1 = run(True)
print(f"{prec}: update phase of one iteration (40 optimiser steps): plain host-enqueue {h0:.1f} ms, wall {w0:.1f} ms | 1-rank RCCL host-enqueue {h1:.1f} ms, wall {w1:.1f} ms", flush=True)
dist.destroy_process_group()

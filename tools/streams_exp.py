"""Experiment: the sections of the update plan (actor + critic merged, discriminator) on one stream vs concurrent streams (timing only; the
shared split-K scratch makes the concurrent results meaningless)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ag = ADDAgent(load_config("train", ["engine.num_envs=4096", f"agent.matmul_precision={prec}"]))
for w in ag._W.values():
    if w.dtype == torch.float32: w.normal_()
for r in (ag._run_actor, ag._run_critic, ag._run_disc):
    for t in r.h + r.dz: t.normal_()
plan = ag._update_plan
marks = [0] + [m for _, m in ag._update_marks]
main = torch.cuda.current_stream()
side = [torch.cuda.Stream() for _ in range(2)]
def seq():
    plan.run(main.cuda_stream)
def par():
    ev = torch.cuda.Event(); ev.record(main)
    streams = [main] + side
    for s in side: s.wait_event(ev)
    for i in range(len(marks) - 1):
        plan.run(streams[i].cuda_stream, marks[i], marks[i + 1])
    for s in side:
        e = torch.cuda.Event(); e.record(s); main.wait_event(e)
for name, fn in (("one stream", seq), ("one stream per section", par), ("one stream", seq), ("one stream per section", par)):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    print(f"{prec} {name}: {(time.perf_counter() - t0) * 100:.3f} ms per optimiser-step plan", flush=True)

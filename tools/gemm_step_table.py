"""Per-launch table of the GEMMs of one optimiser step (N=4096 envs -> 16384-row minibatch): shape, layout, epilogue,
HIP-event time (median of 7), TFLOP/s.  The same launches bench.py times as a whole for roofline.achieved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
ag = ADDAgent(load_config("train", ["engine.num_envs=4096", f"agent.matmul_precision={prec}"] + sys.argv[2:]))
print("matmul_precision:", prec)
for w in ag._W.values():
    if w.dtype == torch.float32: w.normal_()
for r in (ag._run_actor, ag._run_critic, ag._run_disc):
    for t in r.h + r.dz: t.normal_()
    if r.storage16:  # (random bf16 values in every plane: zero operands would flatter the clock)
        for t in r.h16 + r.dz16: t.copy_(torch.randn(t.shape, device=t.device))
for w in ag._W.values():
    if w.dtype == torch.bfloat16: w.copy_(torch.randn(w.shape, device=w.device))
st = torch.cuda.current_stream()
# the clocks of an idle chip ramp over the first milliseconds of load: bring it to its loaded state before the first timed launch
_w = torch.randn(8192, 8192, device="cuda")
for _ in range(60): _w @ _w
torch.cuda.synchronize()
plan = ag._update_plan   # the recorded optimiser step (addhip_plan_t): launches are listed and replayed one by one through the C ABI
calls = [(i, gemms) for i, (name, gemms) in enumerate(plan.launches()) if gemms or name == "addhip_actor_head"]
tot_ms, tot_fl = 0.0, 0.0
for n, (i, gemms) in enumerate(calls):
    if not gemms:  # addhip_actor_head: the head's three 32-wide products + loss in one launch
        ts = []
        for rep in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); plan.run(st.cuda_stream, i, i + 1); e1.record(st); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts[1:])[3]
        fl = 3 * 2.0 * ag.Mb * 32 * ag._model.actor.hidden[-1]
        tot_ms += ms; tot_fl += fl
        print(f"{n:2d} addhip_actor_head (head forward + loss + dWh + dz, one launch)                {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF  {fl/1e9:6.1f} GF", flush=True)
        continue
    cnt, g = len(gemms), gemms[0]   # grouped launch: several equal-shaped problems
    ts = []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); plan.run(st.cuda_stream, i, i + 1); e1.record(st); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts[1:])[3]
    fl = 2.0 * g.M * g.N * g.K * cnt
    tot_ms += ms; tot_fl += fl
    print(f"{n:2d} M={g.M:6d} N={g.N:5d} K={g.K:6d} akc={g.a_kcontig} bkc={g.b_kcontig} epi={g.epilogue} split={g.split_k:2d} x{cnt} "
          f"{ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF  {fl/1e9:6.1f} GF", flush=True)
print(f"sum of isolated launches: {tot_ms:.3f} ms, {tot_fl/1e9:.1f} GFLOP, {tot_fl/tot_ms/1e9:.1f} TFLOP/s")

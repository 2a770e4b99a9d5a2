"""Phase-by-phase timing of one iteration with a sync after every phase (diagnostic; writes progressively)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = open(os.path.join("gpurun_out", "diag.log"), "a")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
t0 = time.time()
import torch
say("import torch", time.time() - t0)
import add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
import add_gym_amd._lib as L
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = load_config("train", [f"engine.num_envs={N}"])
t = time.time(); ag = ADDAgent(cfg); torch.cuda.synchronize(); say("agent built", time.time() - t)
t = time.time(); ag.reset_all_envs(); ag._init_train(); torch.cuda.synchronize(); say("reset", time.time() - t)
for it in range(2):
    t = time.time(); ag._B["ep_stats"].zero_()
    for s in range(ag.T):
        ag._decide_action(s, s, False)
        if s == 0: torch.cuda.synchronize(); say("  decide0", time.time() - t)
        ag._step_env(s, ag._step_out[s], ag._env_c)
        if s == 0: torch.cuda.synchronize(); say("  step0", time.time() - t)
        ag._reset_envs(False, ag._B["obs"][s + 1], ag._B["disc_obs"][s + 1], ag._B["disc_demo"][s + 1], s * 2 + 1)
        if s == 0: torch.cuda.synchronize(); say("  reset0", time.time() - t)
    ag._total_samples += ag.T * ag.N
    torch.cuda.synchronize(); say("rollout", time.time() - t)
    t = time.time(); ag._build_train_data(); torch.cuda.synchronize(); say("build_train_data", time.time() - t)
    t = time.time()
    ag._next_minibatch_indices(); L.call("addhip_gather_minibatch", ag._gather_c, L.current_stream()); torch.cuda.synchronize(); say("  gather", time.time() - t)
    for i, (name, fn, args) in enumerate(ag._update_plan.calls):
        t1 = time.time(); rc = fn(*args, L.current_stream()); torch.cuda.synchronize()
        if it == 1 or time.time() - t1 > 0.05: say("   call", i, name, rc, round((time.time() - t1) * 1e3, 3), "ms")
    say("update step", time.time() - t)
    t = time.time(); ag._update_model(); torch.cuda.synchronize(); say("update_model", time.time() - t)
    t = time.time(); ag._update_normalizers(); torch.cuda.synchronize(); say("normalizers", time.time() - t)
say("done", time.time() - t0)

"""Per-tensor gradient error of one full-size optimiser step (4096 envs, 16384-row minibatch) under each matmul precision,
against the SAME minibatch evaluated in float64 on the CPU, next to torch-CPU's own fp32 autograd error.  Evidence for the
tolerances of tests/test_hip_fullsize.py (profiles/r02_grad_error_table.log)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import learn as OL
from tests.test_hip_agent import make_cfg
from tests.test_hip_fullsize import _fill_minibatch
import add_gym_amd.learning.add_agent as A

torch.set_num_threads(16)
seed = 11
params = OL.synth_params(seed)
model32 = OL.Model(params)
res = {}
mb = None
for prec in sys.argv[1:] or ["fp32", "bf16x3", "bf16x2", "bf16"]:
    cfg = make_cfg(4096, steps_per_iter=32, matmul_precision=prec)
    cfg["task"]["motion_file"] = "synthetic:1x300"
    ag = A.ADDAgent(cfg)
    ag._model.load({k: torch.tensor(v) for k, v in params.items()})
    if hasattr(ag._model, "refresh_shadow"):
        ag._model.refresh_shadow()
    mb = _fill_minibatch(ag, model32, 5)
    ag._W["stats"].zero_()
    ag._run_update_sections()
    torch.cuda.synchronize()
    m = ag._model
    res[prec] = {k: v.numpy().astype(np.float64) for k, v in m.export(m.grads).items() if k != "_model._action_dist._logstd_net"}
    del ag
names = model32.names()
loss, _ = OL.compute_loss(model32, OL.LossCfg(), mb)
g32 = {n: g.numpy().astype(np.float64) for n, g in zip(names, torch.autograd.grad(loss, [model32.p[n] for n in names]))}
with OL.float64_mode():
    model64 = OL.Model(params)
    loss, _ = OL.compute_loss(model64, OL.LossCfg(), mb)
    g64 = {n: g.numpy() for n, g in zip(names, torch.autograd.grad(loss, [model64.p[n] for n in names]))}
print("max |g - g64| / max|g64| per tensor   (and relative L2 error)")
print(f"{'tensor':44s} {'torch-cpu fp32':>22s} " + " ".join(f"{p:>22s}" for p in res))
for n in names:
    s = np.abs(g64[n]).max() + 1e-300
    l2 = np.linalg.norm(g64[n]) + 1e-300
    row = [f"{np.abs(g32[n]-g64[n]).max()/s:9.2e} ({np.linalg.norm(g32[n]-g64[n])/l2:8.2e})"]
    for p in res:
        row.append(f"{np.abs(res[p][n]-g64[n]).max()/s:9.2e} ({np.linalg.norm(res[p][n]-g64[n])/l2:8.2e})")
    print(f"{n:44s} " + " ".join(f"{r:>22s}" for r in row))

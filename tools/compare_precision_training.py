"""Same seeds, same clip, N training iterations under agent.matmul_precision=fp32 and =bf16x3: per-iteration log
scalars side by side (evidence that the exact 3-way bf16 split trains like the fp32 MFMA; the runs still separate slowly,
as any two fp32 summation orders do in a chaotic optimisation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
keys = ("loss", "actor_loss", "critic_loss", "disc_loss", "disc_grad_penalty", "disc_pos_acc", "disc_neg_acc", "disc_reward_mean", "adv_std", "clip_frac", "mean_return")
runs = {}
modes = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("fp32", "bf16x3")
for prec in modes:
    torch.manual_seed(0)
    ag = ADDAgent(load_config("train", [f"engine.num_envs={N}", f"agent.matmul_precision={prec}", "task.motion_file=synthetic:2x600", "seed=3"]))
    ag.reset_all_envs(); ag._init_train()
    rows = []
    for it in range(iters):
        info = ag._train_iter(); ag._iter += 1
        rows.append([float(info[k]) for k in keys])
    runs[prec] = rows
    pn = float(ag._model.params.double().norm())
    print(f"{prec}: |params| after {iters} iterations = {pn:.6f}", flush=True)
    del ag
print("iter  " + "  ".join(f"{k[:14]:>14s}" for k in keys))
for it in range(iters):
    for prec in modes:
        print(f"{it:3d} {prec:6s}" + "  ".join(f"{v:14.6f}" for v in runs[prec][it]))
for other in modes[1:]:
    worst = max(abs(a - b) / (abs(a) + abs(b) + 1e-6) for ra, rb in zip(runs[modes[0]], runs[other]) for a, b in zip(ra, rb))
    print(f"{modes[0]} vs {other}: largest symmetric relative difference of any logged scalar over {iters} iterations: {worst:.3e}")

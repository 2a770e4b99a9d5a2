"""A short real training run through the public surface (ADDAgent.train_model): N iterations at 4096 envs on the synthetic
clips with the kinematic stand-in engine; prints the log.txt columns that show the policy, critic and discriminator moving."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
engine = sys.argv[3] if len(sys.argv) > 3 else "kinematic"
motion = sys.argv[4] if len(sys.argv) > 4 else "synthetic:3x900"
N, T = 4096, 32
print(f"engine {engine}, motion {motion}")
cfg = load_config("train", [f"engine={engine}", f"engine.num_envs={N}", f"agent.matmul_precision={prec}", f"task.motion_file={motion}", "agent.iters_per_output=10",
                            "agent.test_episodes=0", f"agent.max_samples={iters * N * T}"])
d = tempfile.mkdtemp()
ag = ADDAgent(cfg)
ag.train_model(os.path.join(d, "model.pt"), d, os.path.join(d, "log.txt"))
rows = [l.split() for l in open(os.path.join(d, "log.txt")).read().splitlines()]
hdr, rows = rows[0], rows[1:]
keep = ["Iteration", "Samples", "Train_Return", "Train_Episode_Length", "Critic_Loss", "Actor_Loss", "Disc_Loss", "Disc_Neg_Acc", "Disc_Reward_Mean", "Adv_Std", "Clip_Frac"]
idx = [hdr.index(k) for k in keep]
print(f"precision {prec}: {len(rows)} logged iterations, wall {float(rows[-1][hdr.index('Wall_Time')]):.4f} h")
print("  ".join(f"{k[:14]:>14s}" for k in keep))
for r in rows[:: max(1, len(rows) // 15)] + [rows[-1]]:
    print("  ".join(f"{float(r[i]):14.5g}" for i in idx))
assert all(torch.isfinite(torch.tensor([float(v) for v in r])).all() for r in rows)

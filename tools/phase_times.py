"""Wall time of the phases of one training iteration (sync after each phase), median of 3 iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
from add_gym_amd.config import load_config
from add_gym_amd.learning.add_agent import ADDAgent
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
engine = sys.argv[3] if len(sys.argv) > 3 else "kinematic"
graph = sys.argv[4] if len(sys.argv) > 4 else "false"
extra = sys.argv[5:]   # further config overrides, e.g. agent.rollout_graph=true
ag = ADDAgent(load_config("train", [f"engine={engine}", f"engine.num_envs={N}", f"agent.matmul_precision={prec}", "task.motion_file=synthetic:1x3600",
                                    f"agent.rollout_graph={graph}"] + extra))
ag.reset_all_envs(); ag._init_train()
import gc; gc.collect(); gc.freeze()
acc = {}
def timed(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    acc.setdefault(name, []).append((time.perf_counter() - t) * 1e3)
for it in range(8):
    if it: ag._B["obs"][0].copy_(ag._B["obs"][ag.T])
    timed("rollout", ag._rollout_train)
    timed("build_train_data", ag._build_train_data)
    timed("update_model", ag._update_model)
    timed("normalizers+info", lambda: (ag._update_normalizers(), ag._collect_info(40)))
    ag._iter += 1
print(prec, N, engine, "rollout_graph=" + graph, *extra, {k: round(sorted(v[4:])[len(v[4:]) // 2], 2) for k, v in acc.items()}, "ms", flush=True)

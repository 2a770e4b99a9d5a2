"""configs[2] (16 384 envs, bf16 storage): whole iterations with agent.rollout_precision = bf16x2 (default) and bf16.

    python tools/rollout_precision_cost.py      (GPU box)
"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import add_gym_amd  # noqa: E402,F401
from add_gym_amd.config import load_config  # noqa: E402
from add_gym_amd.learning.add_agent import ADDAgent  # noqa: E402


def iteration_ms(envs, rp, iters=4, warm=2):
    cfg = load_config("train", ["engine=kinematic", f"engine.num_envs={envs}", "task.motion_file=synthetic:5x1200", "agent.matmul_precision=bf16",
                                f"agent.rollout_precision={rp}"])
    ag = ADDAgent(cfg, distributed=False)
    ag.reset_all_envs()
    ag._init_train()
    gc.collect()
    for _ in range(warm):
        ag._train_iter()
        ag._iter += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        ag._train_iter()
        ag._iter += 1
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    n = ag.T * ag.N
    del ag
    gc.collect()
    torch.cuda.empty_cache()
    return ms, n


if __name__ == "__main__":
    for envs in (4096, 16384):
        for rp in ("bf16x2", "bf16", "bf16_storage"):
            ms, n = iteration_ms(envs, rp)
            print(f"envs {envs:6d} rollout_precision {rp:7s} iteration {ms:7.1f} ms  {n / ms * 1e3 / 1e6:6.3f} M env-steps/s", flush=True)

"""What agent.deterministic (fixed-order reductions behind every gradient) costs: whole training iterations at BASELINE configs[1]
(4096 envs, 32 steps, 40 optimiser steps) with the switch off and on, per matmul mode.

    python tools/deterministic_cost.py [precision ...]      (GPU box)
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


def iteration_ms(precision, det, iters=6, warm=2):
    cfg = load_config("train", ["engine=kinematic", "engine.num_envs=4096", "task.motion_file=synthetic:1x3600", f"agent.matmul_precision={precision}",
                                f"agent.deterministic={'true' if det else 'false'}"])
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
    del ag
    gc.collect()
    torch.cuda.empty_cache()
    return ms


if __name__ == "__main__":
    if os.environ.get("DET_ONLY"):  # one setting only, a few iterations: for a kernel trace (rocprofv3 --kernel-trace --stats)
        print(iteration_ms(sys.argv[1], os.environ["DET_ONLY"] == "1", iters=3, warm=1))
        sys.exit(0)
    for prec in sys.argv[1:] or ["fp32", "f16x2", "bf16"]:
        off, on = iteration_ms(prec, False), iteration_ms(prec, True)
        print(f"{prec:8s} iteration {off:7.1f} ms -> deterministic {on:7.1f} ms  (x{on / off:.3f}; {(on - off) / 40 * 1e3:6.1f} us per optimiser step)", flush=True)

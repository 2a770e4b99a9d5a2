#!/usr/bin/env python3
"""Reference-Python CPU baseline: the IMPORTED reference (rsamf/add-gym at /root/reference, unmodified)
driven through tools/ref_harness.py -- stand-ins for its absent third-party imports, FakeEngine (a kinematic stand-in behind the
reference's own engine API, so PHYSICS COST IS EXCLUDED) -- timed on BaseAgent._train_iter only (no test rollouts,
base_agent.py:92-93 excluded): 1 warm-up + 3 timed iterations, median.  Build-container tooling: the reference cannot travel to
the GPU box, so this number is quoted in DESIGN.md beside the oracle's ("reported baseline, not the optimisation target").

    python tools/time_reference.py [num_envs=4096] [timed_iters=3]
"""
import json
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.install_stubs()
import gen_golden_agent as G  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    ag, cfg = G.build_agent(n, clip_frames=None, seed=0)  # the full walk1_subject1_trimmed clip (3726 frames)
    ag._curr_obs, ag._curr_info = ag._reset_envs()
    ag._init_train() if hasattr(ag, "_init_train") else None
    T = cfg["agent"]["steps_per_iter"]
    times = []
    for i in range(1 + iters):
        t0 = time.perf_counter()
        ag._train_iter()
        dt = time.perf_counter() - t0
        print(f"iter {i}{' (warm-up)' if i == 0 else ''}: {dt:.2f} s  -> {T * n / dt:.0f} env-steps/s", flush=True)
        if i > 0:
            times.append(dt)
    med = statistics.median(times)
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
    out = dict(kind="reference", num_envs=n, steps_per_iter=T, update_epochs=cfg["agent"]["update_epochs"], batch_size=cfg["agent"]["batch_size"],
               timed_iters=iters, median_s=med, env_steps_per_s=T * n / med, threads=threads, cpu=cpu[0] if cpu else "?",
               torch=torch.__version__, physics="excluded (kinematic FakeEngine behind the reference's engine API)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()

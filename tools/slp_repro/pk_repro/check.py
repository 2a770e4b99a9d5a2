"""Run pk_repro's two builds (SelectionDAG, GlobalISel; built by build.sh in the build container) on the GPU and compare with numpy."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from run_hsaco import launch  # noqa: E402

rng = np.random.RandomState(1)
x = rng.uniform(-1, 1, (256, 18)).astype(np.float32)
X = x.astype(np.float64)
a, b, c, p, q, r = (X[:, 2 * i:2 * i + 2] for i in range(6))
s0, s1, s2, s3, d = X[:, 12], X[:, 13], X[:, 14], X[:, 15], X[:, 16:18]
want = np.stack([a[:, 0] * s0 + c[:, 0] * s1 + b[:, 1] * s2,
                 a[:, 1] * q[:, 1] + c[:, 1] * p[:, 1] + b[:, 1] * r[:, 1],
                 a[:, 0] * p[:, 0] + a[:, 1] * q[:, 0] + b[:, 1] * r[:, 0],
                 b[:, 1] * d[:, 1] - s3 * a[:, 1]], 1)
for name in sys.argv[1:] or ["pk_sdag", "pk_gisel"]:
    got = launch(os.path.join(HERE, name + ".hsaco"), "pk_repro", x, 4)
    err = np.abs(got - want).max(0)
    print(f"{name:10s} worst |error| per output {err}  {'ok' if err.max() < 1e-5 else 'WRONG'}", flush=True)

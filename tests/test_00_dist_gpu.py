"""The data-parallel exchange path of the PRODUCT on the GPU box (the semantics the reference intends at
base_agent.py:47-57 and normalizer.py:41-58):

* a 1-rank RCCL ("nccl") group drives the asynchronous four-bucket all-reduce on three streams and must reproduce the
  non-distributed optimiser step;
* two ranks sharing the GPU over gloo run the agent end to end: rank 0's initial weights everywhere, post-exchange gradient
  == sum of the ranks' pre-exchange gradients == (1/world folded into the loss coefficients) the mean of the ranks' full
  oracle gradients, parameters identical on both ranks after a whole iteration, normaliser statistics all-reduced.

Each rank is a fresh child process (tests/dist_child.py).  This file sorts first on purpose: the children are started before
this pytest process has made its first GPU call (a process that holds the GPU must not spawn-and-exec on this pool).
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "dist_child.py")


def _gpu_untouched():
    import torch

    return not torch.cuda.is_initialized()


def _env(**extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({k: str(v) for k, v in extra.items()})
    return env


@pytest.mark.timeout(600)
def test_one_rank_rccl_group_takes_the_bucketed_exchange_path(tmp_path):
    if not _gpu_untouched():
        pytest.skip("this process already holds the GPU: run tests/test_00_dist_gpu.py first (it does, in directory order)")
    out = tmp_path / "nccl1.json"
    r = subprocess.run([sys.executable, CHILD, "nccl1", str(out)], env=_env(MASTER_ADDR="127.0.0.1", MASTER_PORT=29731), cwd=ROOT,
                       capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["backend"] == "nccl"
    assert res["weights_bit_equal"], res
    assert res["bias_max_rel_diff"] <= 1e-5, res
    assert all(res["nonzero_buckets"].values()), res
    assert res["train_iter_finite"], res


@pytest.mark.timeout(900)
def test_two_ranks_share_the_gpu_over_gloo_agent_level(tmp_path):
    if not _gpu_untouched():
        pytest.skip("this process already holds the GPU: run tests/test_00_dist_gpu.py first (it does, in directory order)")
    out = tmp_path / "gloo2.json"
    port = 29800 + os.getpid() % 100
    procs = [subprocess.Popen([sys.executable, CHILD, "gloo2", str(out)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=_env(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=rank, WORLD_SIZE=2, LOCAL_RANK=rank))
             for rank in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=800)
        except subprocess.TimeoutExpired:
            p.kill()  # exactly the process started above
            o, _ = p.communicate()
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    res = json.loads(out.read_text())
    assert res["init_params_equal"], res
    assert res["post_equals_sum_of_pre"], res
    assert res["max_err_per_tensor"] <= 2e-4, res          # same bound as the single-rank gradient parity tests
    assert res["params_equal_after_iter"] and res["params_moved"], res
    assert res["obs_norm_equal"] and res["obs_norm_count"] == res["expected_count"], res
    assert res["finite"], res


@pytest.mark.timeout(900)
def test_bench_gpus_2_starts_two_ranks_itself(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it — the shape of the driver's N-GPU command — must itself start one
    process per rank (the reference's launcher: sagemaker-entrypoint.sh:139-147 -> add_gym/main.py:128-176) and report the group it
    ran in.  One GPU here, so the two ranks share it under the gloo rehearsal switch; without the switch the launcher must refuse."""
    if not _gpu_untouched():
        pytest.skip("this process already holds the GPU: run tests/test_00_dist_gpu.py first (it does, in directory order)")
    import torch

    base = {k: v for k, v in _env().items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ADDHIP_DIST_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "256", "--steps", "1", "--warmup", "1", "--no-alt", "--no-cpu-baseline"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, env=base, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "one process per GPU" in r.stderr, r.stderr[-2000:]
    r = subprocess.run(cmd, env=dict(base, ADDHIP_DIST_BACKEND="gloo"), cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["dist"]["group_size"] == 2 and res["dist"]["backend"] == "gloo", res
    assert res["dist"]["launcher"] == "bench.py --gpus"
    assert res["config"]["envs_per_gpu"] == 256 and res["scaling"] == "weak"
    assert abs(res["value"] - 2 * 256 * res["config"]["steps_per_iter"] / (res["ms_per_step"] * 1e-3)) <= 1e-6 * res["value"]
    assert "REHEARSAL" in res["data"] and "roofline" in res

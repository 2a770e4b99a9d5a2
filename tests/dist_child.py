"""Child process of tests/test_00_dist_gpu.py (one process per rank; the parent starts it before its own first GPU call).

    python tests/dist_child.py nccl1 <out.json>       1-rank RCCL group: the asynchronous bucketed exchange path
    python tests/dist_child.py gloo2 <out.json>       RANK/WORLD_SIZE/MASTER_* in the env: 2 ranks share the GPU, gloo backend

Checks are made here (they need both the device buffers and the process group); the verdict goes to <out.json> of rank 0.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F = np.float32


def build_agent(num_envs, steps, seed, distributed):
    import add_gym_amd  # noqa: F401
    from add_gym_amd.config import load_config
    from add_gym_amd.learning.add_agent import ADDAgent
    from tests.util import kin_meta

    cfg = load_config("train", [f"engine.num_envs={num_envs}", f"agent.steps_per_iter={steps}", f"seed={seed}", "task.motion_file=synthetic:2x240"])
    cfg["task"]["motion_joint_order"] = kin_meta()["motion_joint_order"]
    return ADDAgent(cfg, distributed=distributed)


def minibatch_of(ag):
    """The gathered minibatch the update plan is about to consume, as the oracle's dict (host arrays)."""
    W, M = ag._W, ag.Mb
    c = lambda t: t.detach().cpu().numpy().copy()
    return dict(norm_obs=c(W["norm_obs"][:, :264]), norm_action=c(W["norm_act"][:, :29]), a_logp=c(W["mb_logp"]), adv=c(W["mb_adv"]),
                tar_val=c(W["mb_tar"]), rand_action_mask=c(W["mb_mask"]), norm_diff=c(W["norm_diff"][:M, :114]))


def oracle_grad_flat(ag, params, mb):
    """Gradient of the whole loss on `mb` by CPU autograd, laid out like the agent's flat gradient buffer."""
    import torch
    from oracle import learn as OL

    model = OL.Model(params)
    loss, _ = OL.compute_loss(model, OL.LossCfg(), mb)
    names = model.names()
    gs = torch.autograd.grad(loss, [model.p[n] for n in names])
    m = ag._model
    sd = {n: g for n, g in zip(names, gs)}
    # Model.load writes through views of a buffer on the model's device: stage on the device, read back
    dev = torch.zeros(m.count, device=m.params.device)
    m.load(sd, dev)
    return dev.cpu()


def run_nccl1(out_path):
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)  # "nccl" is RCCL on ROCm
    ag = build_agent(1024, 8, 5, True)
    assert ag._distributed and ag._world == 1
    ag.reset_all_envs()
    ag._init_train()
    ag._rollout_train()
    ag._build_train_data()
    ag._next_minibatch_indices()
    import add_gym_amd._lib as L

    L.call("addhip_gather_minibatch", ag._gather_c, ag._stream())
    m = ag._model
    res = {}
    # (a) exchange path: four asynchronous RCCL buckets on three streams
    ag._run_update_sections()
    torch.cuda.synchronize()
    g_ex = m.grads.clone()
    # (b) the same step without a process group in the picture
    ag._distributed = False
    ag._run_update_sections()
    torch.cuda.synchronize()
    g_plain = m.grads.clone()
    ag._distributed = True
    # weight gradients come out of deterministic split-K slab reductions: bit-equal; bias / scalar-head gradients are
    # accumulated by atomics (order varies run to run): equal to fp32 rounding
    br = m.bucket_ranges
    bias = torch.zeros(m.count, dtype=torch.bool, device=g_ex.device)
    for (net, key), (off, shape) in m.offsets.items():
        if key.startswith("b") or (key == "Wh" and net in ("critic", "disc")):
            bias[off:off + int(np.prod(shape))] = True
    res["weights_bit_equal"] = bool(torch.equal(g_ex[~bias], g_plain[~bias]))
    scale = float(g_plain.abs().max())
    res["bias_max_rel_diff"] = float((g_ex[bias] - g_plain[bias]).abs().max()) / scale
    res["grad_abs_max"] = scale
    res["nonzero_buckets"] = {k: bool(g_ex[a:b].abs().sum() > 0) for k, (a, b) in br.items()}
    # a whole optimiser-step loop through the public path still works with the group up
    info = ag._train_iter()
    torch.cuda.synchronize()
    res["train_iter_finite"] = bool(all(np.isfinite(v) for v in info.values()))
    res["backend"] = dist.get_backend()
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def run_gloo2(out_path):
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)  # both ranks share the one GPU of the box (rehearsal of the plumbing, not a measurement)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(4)
    ag = build_agent(64, 4, 3, True)  # agent seed = seed * 1000003 + rank: different rollouts per rank
    m = ag._model
    res = {"rank": rank}
    # DDP-constructor semantics: every rank starts from rank 0's weights
    p0 = m.params.clone()
    dist.broadcast(p0, 0)
    res["init_params_equal"] = bool(torch.equal(p0, m.params))
    ag.reset_all_envs()
    ag._init_train()
    ag._rollout_train()
    ag._build_train_data()
    ag._next_minibatch_indices()
    import add_gym_amd._lib as L
    from add_gym_amd import dist as D

    L.call("addhip_gather_minibatch", ag._gather_c, ag._stream())
    torch.cuda.synchronize()
    mb = minibatch_of(ag)
    params = {k: v.numpy() for k, v in m.export().items() if k != "_model._action_dist._logstd_net"}
    # capture the gradient as it is just before the exchange
    seen = {}
    orig = D.all_reduce_sum_

    def spy(*tensors):
        if len(tensors) == 1 and tensors[0].data_ptr() == m.grads.data_ptr():
            torch.cuda.synchronize()
            seen["pre"] = tensors[0].clone()
        return orig(*tensors)

    D.all_reduce_sum_ = spy
    ag._run_update_sections()
    D.all_reduce_sum_ = orig
    torch.cuda.synchronize()
    post = m.grads.clone()
    pre = seen["pre"].cpu()  # (gloo gathers host tensors only)
    both = [torch.zeros_like(pre) for _ in range(world)]
    dist.all_gather(both, pre)
    res["post_equals_sum_of_pre"] = bool(torch.equal(post.cpu(), both[0] + both[1]))
    # every loss coefficient of the plan carries 1/world: the SUM over ranks must be the gradient of the mean loss, i.e. the
    # mean over ranks of each rank's full (unscaled) oracle gradient
    og = oracle_grad_flat(ag, params, mb)
    dist.all_reduce(og)
    og /= world
    scale = float(og.abs().max())
    res["max_err_vs_oracle_mean_grad"] = float((post.cpu() - og).abs().max()) / scale
    worst = 0.0
    for (net, key), (off, shape) in m.offsets.items():
        n = int(np.prod(shape))
        s = float(og[off:off + n].abs().max()) + 1e-12
        worst = max(worst, float((post.cpu()[off:off + n] - og[off:off + n]).abs().max()) / s)
    res["max_err_per_tensor"] = worst
    # one whole iteration: parameters stay identical across ranks, normaliser statistics agree
    ag._iter_started = False
    ag._total_samples = 0
    ag._perm = None
    info = ag._train_iter()
    torch.cuda.synchronize()
    gp = [torch.zeros(m.count) for _ in range(world)]
    dist.all_gather(gp, m.params.cpu())
    res["params_equal_after_iter"] = bool(torch.equal(gp[0], gp[1]))
    res["params_moved"] = bool(not torch.equal(gp[0], p0.cpu()))
    gm = [torch.zeros(ag._Nrm["obs_mean"].shape[0]) for _ in range(world)]
    dist.all_gather(gm, ag._Nrm["obs_mean"].cpu())
    res["obs_norm_equal"] = bool(torch.equal(gm[0], gm[1]))
    res["obs_norm_count"] = int(ag._Nrm["obs_cnt"].item())
    res["expected_count"] = world * ag.T * ag.N
    res["finite"] = bool(all(np.isfinite(v) for v in info.values()))
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    {"nccl1": run_nccl1, "gloo2": run_gloo2}[sys.argv[1]](sys.argv[2])

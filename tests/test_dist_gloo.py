"""world_size-2 data-parallel exchange on CPU (gloo): the mean of the per-rank gradients equals the gradient of the
concatenated minibatch, and all-reduced normaliser sums equal single-process statistics."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_params(seed):
    from oracle import learn as OL

    return OL.synth_params(seed)


def _batch(rng, m):
    return dict(norm_obs=rng.standard_normal((m, 264)).astype(np.float32), norm_action=(rng.standard_normal((m, 29)) * 0.05).astype(np.float32),
                a_logp=rng.standard_normal(m).astype(np.float32), adv=rng.standard_normal(m).astype(np.float32), tar_val=rng.standard_normal(m).astype(np.float32),
                rand_action_mask=np.ones(m, np.float32), norm_diff=rng.standard_normal((m, 114)).astype(np.float32))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import add_gym_amd  # noqa: F401
    from add_gym_amd import dist as D
    from oracle import learn as OL

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    rng = np.random.RandomState(0)
    full = _batch(rng, 64)
    model = OL.Model(_small_params(7))
    # old log-probs consistent with the policy so that ratios are O(1)
    with torch.no_grad():
        full["a_logp"] = model.log_prob(model.actor_mean(OL.t32(full["norm_obs"])), OL.t32(full["norm_action"])).numpy()
    shard = {k: v[rank * 32:(rank + 1) * 32] for k, v in full.items()}
    loss, _ = OL.compute_loss(model, OL.LossCfg(), shard)
    names = model.names()
    grads = torch.autograd.grad(loss, [model.p[n] for n in names])
    flat = torch.cat([g.reshape(-1) for g in grads])
    flat_b = flat.clone()
    D.all_reduce_mean_(flat)
    # the agent's exchange: one asynchronous bucket per net, started in turn, one wait, one scale (add_agent._update_model)
    third = flat_b.numel() // 3
    cuts = [0, third, 2 * third, flat_b.numel()]
    pending = [D.all_reduce_sum_async(flat_b[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    D.wait_all(pending)
    flat_b.mul_(1.0 / world)
    # normaliser sums
    x = torch.tensor(rng.standard_normal((2, 50, 9)).astype(np.float32))[rank]
    s1, s2 = x.sum(0), (x * x).sum(0)
    w = D.all_reduce_sum_(s1, s2)
    b = torch.full((5,), float(rank))
    D.broadcast_(b, 0)
    if rank == 0:
        torch.save(dict(flat=flat, flat_b=flat_b, s1=s1, s2=s2, w=w, b=b), tmp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_mean_and_normaliser_sums(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import learn as OL

    tmp = str(tmp_path / "out.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
    got = torch.load(tmp, weights_only=True)
    rng = np.random.RandomState(0)
    full = _batch(rng, 64)
    model = OL.Model(_small_params(7))
    with torch.no_grad():
        full["a_logp"] = model.log_prob(model.actor_mean(OL.t32(full["norm_obs"])), OL.t32(full["norm_action"])).numpy()
    # single process, concatenated minibatch.  The batch-mean terms (PPO, critic, BCE, gradient penalty) average exactly;
    # the parameter-only L2 terms are identical on every rank, so their mean is themselves.
    loss, _ = OL.compute_loss(model, OL.LossCfg(), full)
    names = model.names()
    ref = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, [model.p[n] for n in names])])
    # the zero-difference "positive" sample enters every rank's loss once: mean over ranks == single-process value as well
    scale = ref.abs().max()
    assert (got["flat"] - ref).abs().max() <= 2e-5 * scale
    assert torch.equal(got["flat_b"], got["flat"])  # bucketed asynchronous exchange == one flat all-reduce
    x = torch.tensor(rng.standard_normal((2, 50, 9)).astype(np.float32)).reshape(100, 9)
    assert got["w"] == 2
    torch.testing.assert_close(got["s1"], x.sum(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got["s2"], (x * x).sum(0), rtol=1e-5, atol=1e-5)
    assert torch.all(got["b"] == 0)

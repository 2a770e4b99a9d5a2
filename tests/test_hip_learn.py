"""Learner-side HIP kernels through the C ABI against the CPU oracle (oracle/learn.py, oracle/task.py)
and the reference's golden vectors."""
import math

import numpy as np
import pytest

from oracle import learn as OL
from oracle import task as OT
from tests.util import gload

pytestmark = pytest.mark.gpu
F = np.float32


_KEEP = []


def P(t):
    """Device address of a tensor that is kept alive until the test module is torn down: launches are
    asynchronous, so a temporary freed right after ptr() could be recycled by the caching allocator
    before the kernel has read it."""
    import add_gym_amd._lib as L

    _KEEP.append(t)
    return L.ptr(t)


def T(x, dtype=None):
    import torch

    return torch.tensor(np.ascontiguousarray(x), dtype=dtype, device="cuda")


def logp_const():
    import torch

    logstd = torch.full((29,), float(np.log(0.05)), dtype=torch.float32)
    return float((-0.5 * 29 * np.log(2.0 * np.pi) - torch.sum(logstd)).item()), float(torch.exp(logstd)[0].item())


@pytest.mark.parametrize("name", ["actor_step", "actor_step_constant_std", "actor_step_variable_std"])
def test_actor_sample_matches_reference_golden(name):
    """(`_constant_std`: actor_std_type CONSTANT -- a trainable log-std per action dimension through addhip_dist_refresh's vector;
    `_variable_std`: VARIABLE -- the per-sample log-std in columns 32..60 of a 64-wide head output.)"""
    import torch
    import add_gym_amd._lib as L
    from tests.test_oracle_vs_golden import golden_logstd

    g = gload(name)
    variable = golden_logstd(name) == "variable"
    params = OL.synth_params(int(g["seed"]), logstd=golden_logstd(name))
    model = OL.Model(params)
    on = OL.Normalizer(264, g["obs_mean"], g["obs_std"])
    with torch.no_grad():
        mean, ls_rows, _ = model.dist(OL.t32(on.normalize(g["obs"])))
        mean = mean.numpy()
    n = mean.shape[0]
    ldm = 64 if variable else 32
    mean32 = np.zeros((n, ldm), F)
    mean32[:, :29] = mean
    if variable:
        mean32[:, 32:61] = ls_rows.numpy()
    c, std = logp_const()
    dist = None
    if OL.LOGSTD_KEY in params:  # the scalars are ignored then
        dist_t = torch.zeros(L.DIST_FLOATS, device="cuda")
        L.call("addhip_dist_refresh", P(T(params[OL.LOGSTD_KEY])), L.ptr(dist_t), L.current_stream())
        torch.cuda.synchronize()
        ls = params[OL.LOGSTD_KEY]
        np.testing.assert_allclose(dist_t.cpu().numpy()[:29], np.exp(ls), rtol=2e-7)
        c = float(dist_t[32])
        np.testing.assert_allclose(c, -0.5 * 29 * np.log(2 * np.pi) - ls.astype(np.float64).sum(), rtol=1e-6)
        np.testing.assert_allclose(float(dist_t[33]), ls.astype(np.float64).sum() + 0.5 * 29 * np.log(2 * np.pi * np.e), rtol=1e-6)
        dist, std = L.ptr(dist_t), float("nan")
    act = torch.zeros(n, 32, device="cuda")
    logp = torch.zeros(n, device="cuda")
    mask = torch.zeros(n, device="cuda")
    dmean = T(mean32)
    rows = L.ptr(dmean) + 4 * 32 if variable else None  # (the scalars / dist are ignored then)
    if variable:
        std, c = float("nan"), float("nan")
    L.call("addhip_actor_sample", L.ptr(dmean), ldm, P(T(g["noise"])), std, c if dist is None else float("nan"), dist, rows, P(T(g["a_mean"])), P(T(g["a_std"])), n, 0, None, 1.0,
           L.ptr(act), L.ptr(logp), L.ptr(mask), L.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(act.cpu().numpy()[:, :29], g["action"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), g["a_logp"], rtol=1e-5, atol=1e-4)
    assert np.all(mask.cpu().numpy() == 1) and np.all(act.cpu().numpy()[:, 29:] == 0)
    # exploration probability < 1 (ppo_agent.py:80-88): env i explores iff u[i] < p, else it takes the mode with mask 0
    u = np.random.default_rng(3).random(n).astype(F)
    act2, logp2, mask2 = torch.zeros(n, 32, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    L.call("addhip_actor_sample", L.ptr(dmean), ldm, P(T(g["noise"])), std, c, dist, rows, P(T(g["a_mean"])), P(T(g["a_std"])), n, 0, P(T(u)), 0.5,
           L.ptr(act2), L.ptr(logp2), L.ptr(mask2), L.current_stream())
    torch.cuda.synchronize()
    ex = u < 0.5
    assert 0 < ex.sum() < n and np.array_equal(mask2.cpu().numpy(), ex.astype(F))
    assert torch.equal(act2[T(ex)], act[T(ex)]) and torch.equal(logp2[T(ex)], logp[T(ex)])
    mode = (mean * g["a_std"] + g["a_mean"]).astype(F)
    np.testing.assert_allclose(act2.cpu().numpy()[~ex][:, :29], mode[~ex], rtol=0, atol=1e-6)
    if variable:  # log-density at the mode: the row's own constant
        c = (-0.5 * 29 * np.log(2 * np.pi) - ls_rows.numpy().astype(np.float64).sum(1))[~ex]
    np.testing.assert_allclose(logp2.cpu().numpy()[~ex], c, rtol=2e-6)  # log-density at the mode


def test_td_lambda_adv_matches_reference_golden():
    import torch
    import add_gym_amd._lib as L

    g = gload("td_lambda_adv")
    Tn, n = g["r"].shape
    nv = T(g["next_vals"])
    tar = torch.zeros(Tn, n, device="cuda")
    adv = torch.zeros(Tn, n, device="cuda")
    scratch = torch.zeros(4096, dtype=torch.float64, device="cuda")
    stats = torch.zeros(2, device="cuda")
    L.call("addhip_td_lambda_adv", P(T(g["r"])), L.ptr(nv), None, P(T(g["vals"])), P(T(g["done"], torch.int32)), P(torch.ones(Tn, n, device="cuda")),
           Tn, n, 0.99, 0.95, 0.0, 0.0, 4.0, L.ptr(tar), L.ptr(adv), L.ptr(scratch), L.ptr(stats), L.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(tar.cpu().numpy(), g["tar_val"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(stats.cpu().numpy(), [float(g["adv_mean"]), float(g["adv_std"])], rtol=1e-5)
    np.testing.assert_allclose(adv.cpu().numpy(), g["adv"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(nv.cpu().numpy(), g["next_vals"])  # read-only input


def test_td_lambda_shifted_values_and_timeout_rows():
    """The agent's layout: next_vals aliases vals shifted by one slot, and the value of the true next obs of a DONE_TIME
    sample comes from timeout_vals[env] (its obs[t+1] row already holds the reset obs).  Same returns as the reference
    recurrence on an explicit next_vals array (oracle restatement of base_agent.py:624-647)."""
    import torch
    import add_gym_amd._lib as L
    from oracle import learn as OL

    g = gload("td_lambda_adv")
    Tn, n = g["r"].shape
    rng = np.random.default_rng(5)
    done = g["done"].copy()
    for e in range(n):  # at most one DONE_TIME per env and call
        ts = np.nonzero(done[:, e] == 3)[0]
        done[ts[:-1], e] = 0
    assert (done == 3).sum() > 0
    vals_all = rng.normal(size=(Tn + 1, n)).astype(F)      # V(obs[0..T])
    tv = rng.normal(size=n).astype(F)                       # V(pre-reset obs) of each env's DONE_TIME sample
    next_ref = vals_all[1:].copy()
    tsel = done == 3
    next_ref[tsel] = np.broadcast_to(tv, (Tn, n))[tsel]
    next_ref[(done == 1) | (done == 2)] = 0.0
    want = OL.td_lambda_return(g["r"], next_ref, done, 0.99, 0.95)
    va = T(vals_all)
    tar = torch.zeros(Tn, n, device="cuda")
    adv = torch.zeros(Tn, n, device="cuda")
    scratch = torch.zeros(4096, dtype=torch.float64, device="cuda")
    stats = torch.zeros(2, device="cuda")
    L.call("addhip_td_lambda_adv", P(T(g["r"])), L.ptr(va[1:]), P(T(tv)), L.ptr(va), P(T(done, torch.int32)), P(torch.ones(Tn, n, device="cuda")),
           Tn, n, 0.99, 0.95, 0.0, 0.0, 4.0, L.ptr(tar), L.ptr(adv), L.ptr(scratch), L.ptr(stats), L.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(tar.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    assert np.array_equal(va.cpu().numpy(), vals_all)


def test_normalizers_match_reference_golden():
    import torch
    import add_gym_amd._lib as L

    g = gload("normalizers")
    mean, std, msq = torch.zeros(7, device="cuda"), torch.ones(7, device="cuda"), torch.zeros(7, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    s1, s2 = torch.zeros(7, device="cuda"), torch.zeros(7, device="cuda")
    ma, dcnt, sa = torch.ones(5, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(5, device="cuda")
    for it in range(3):
        for s in range(4):
            x, y = T(g[f"x{it}_{s}"]), T(g[f"y{it}_{s}"])
            L.call("addhip_norm_accum", L.ptr(x), 33, 7, 7, L.ptr(s1), L.ptr(s2), L.current_stream())
            L.call("addhip_norm_accum", P(torch.abs(y).contiguous()), 33, 5, 5, L.ptr(sa), None, L.current_stream())
        L.call("addhip_norm_merge", L.ptr(mean), L.ptr(std), L.ptr(msq), L.ptr(cnt), L.ptr(s1), L.ptr(s2), 132, 7, 1e-8, int(it == 0), L.current_stream())
        L.call("addhip_diffnorm_merge", L.ptr(ma), L.ptr(dcnt), L.ptr(sa), 132, 5, L.current_stream())
        torch.cuda.synchronize()
        np.testing.assert_allclose(mean.cpu().numpy(), g[f"mean{it}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(std.cpu().numpy(), g[f"std{it}"], rtol=1e-5)
        np.testing.assert_allclose(ma.cpu().numpy(), g[f"mean_abs{it}"], rtol=1e-5)
        assert int(cnt.item()) == int(g[f"count{it}"][0]) and int(dcnt.item()) == int(g[f"dcount{it}"][0])


def test_disc_prep_sampler_and_reward():
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(5)
    rows, dim, stride = 6000, 114, 116
    a = np.zeros((rows, stride), F)
    d = np.zeros((rows, stride), F)
    a[:, :dim] = rng.standard_normal((rows, dim))
    d[:, :dim] = rng.standard_normal((rows, dim))
    mean_abs = (rng.rand(stride) * 0.5 + 1e-5).astype(F)
    lengths = np.asarray([6.6333, 2.0, 9.3], F)
    ids = rng.randint(0, 3, rows)
    times = (rng.rand(rows).astype(F) * lengths[ids] * 1.05).astype(F)
    smp = OT.SegmentSampler(lengths, 0.01, 20, None, 0.02)
    smp.errors = (rng.rand(3, 20) * 2).astype(F)
    err0 = smp.errors.copy()
    t = dict(err=T(err0), seg=T(smp.segment_sizes), cdf=T(np.asarray([0.3, 0.6, 1.0], F)), bits=torch.zeros(1, dtype=torch.int32, device="cuda"),
             es=torch.zeros(60, device="cuda"), ec=torch.zeros(60, device="cuda"))
    sc = L.SamplerT(L.ptr(t["err"]), L.ptr(t["seg"]), L.ptr(t["cdf"]), 20, -1.0, 0.02, 1, L.ptr(t["bits"]), L.ptr(t["es"]), L.ptr(t["ec"]))
    nd = torch.full((rows, stride), 3.0, device="cuda")
    abs_sum = torch.zeros(stride, device="cuda")
    L.call("addhip_disc_prep", P(T(a)), P(T(d)), stride, dim, rows, P(T(mean_abs)), 1e-4, L.ptr(nd), P(T(ids, torch.int32)), P(T(times)),
           sc, 3, L.ptr(abs_sum), L.current_stream())
    L.call("addhip_sampler_update", sc, 3, L.current_stream())
    torch.cuda.synchronize()
    dn = OL.DiffNormalizer(dim)
    dn.mean_abs = mean_abs[:dim]
    np.testing.assert_allclose(nd.cpu().numpy()[:, :dim], dn.normalize(d[:, :dim] - a[:, :dim]), rtol=1e-6, atol=1e-6)
    assert np.all(nd.cpu().numpy()[:, dim:] == 0)
    np.testing.assert_allclose(abs_sum.cpu().numpy()[:dim], np.abs(d - a)[:, :dim].astype(np.float64).sum(0), rtol=1e-5)
    diff = a[:, :dim] - d[:, :dim]
    smp.update_errors(ids, times, np.sum(diff * diff, axis=-1, dtype=F))
    np.testing.assert_allclose(t["err"].cpu().numpy().reshape(3, 20), smp.errors, rtol=1e-5)
    assert float(t["es"].abs().sum()) == 0 and float(t["ec"].abs().sum()) == 0
    # disc reward (amp_agent.py:194-206) + reward mix
    logits = (rng.standard_normal(rows) * 4).astype(F)
    task_r = rng.rand(rows).astype(F)
    rew = T(task_r)
    stats = torch.zeros(2, device="cuda")
    L.call("addhip_disc_reward", P(T(logits)), L.ptr(rew), rows, 2.0, 0.0, 1.0, L.ptr(stats), L.current_stream())
    torch.cuda.synchronize()
    dr = OL.disc_reward(logits, 2.0)
    np.testing.assert_allclose(rew.cpu().numpy(), dr, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(stats.cpu().numpy()[0], dr.astype(np.float64).sum(), rtol=1e-5)


def test_gather_minibatch():
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(6)
    R, Mb = 3000, 777
    obs = rng.standard_normal((R, 264)).astype(F)
    act = np.zeros((R, 32), F)
    act[:, :29] = rng.standard_normal((R, 29))
    do, dd = np.zeros((R, 116), F), np.zeros((R, 116), F)
    do[:, :114], dd[:, :114] = rng.standard_normal((R, 114)), rng.standard_normal((R, 114))
    sc = {k: rng.standard_normal(R).astype(F) for k in ("logp", "adv", "tar", "mask")}
    om, os_ = rng.standard_normal(264).astype(F), (rng.rand(264) + 0.5).astype(F)
    am, as_ = rng.standard_normal(29).astype(F), (rng.rand(29) + 0.5).astype(F)
    ma = (rng.rand(116) * 0.3).astype(F)
    idx = rng.randint(0, R, Mb).astype(np.int64)
    d = {k: T(v) for k, v in dict(obs=obs, act=act, do=do, dd=dd, om=om, os=os_, am=am, as_=as_, ma=ma, idx=idx, **sc).items()}
    o = dict(no=torch.zeros(Mb, 264, device="cuda"), na=torch.ones(Mb, 32, device="cuda"), lp=torch.zeros(Mb, device="cuda"), ad=torch.zeros(Mb, device="cuda"),
             tv=torch.zeros(Mb, device="cuda"), mk=torch.zeros(Mb, device="cuda"), nd=torch.ones(Mb, 116, device="cuda"))
    g = L.GatherT(L.ptr(d["idx"]), Mb, L.ptr(d["obs"]), 264, 264, L.ptr(d["om"]), L.ptr(d["os"]), L.ptr(d["act"]), L.ptr(d["am"]), L.ptr(d["as_"]),
                  L.ptr(d["logp"]), L.ptr(d["adv"]), L.ptr(d["tar"]), L.ptr(d["mask"]), L.ptr(d["do"]), L.ptr(d["dd"]), 116, 114, L.ptr(d["ma"]), 1e-4,
                  L.ptr(o["no"]), L.ptr(o["na"]), L.ptr(o["lp"]), L.ptr(o["ad"]), L.ptr(o["tv"]), L.ptr(o["mk"]), L.ptr(o["nd"]))
    L.call("addhip_gather_minibatch", g, L.current_stream())
    torch.cuda.synchronize()
    # optional bf16 copies (bf16-storage mode): exactly the fp32 outputs rounded to nearest even
    fp32_out = {k: o[k].clone() for k in ("no", "nd")}
    o16 = dict(no=torch.ones(Mb, 264, device="cuda", dtype=torch.bfloat16), nd=torch.ones(Mb, 116, device="cuda", dtype=torch.bfloat16))
    g.norm_obs16, g.norm_diff16 = L.ptr(o16["no"]), L.ptr(o16["nd"])
    L.call("addhip_gather_minibatch", g, L.current_stream())
    torch.cuda.synchronize()
    for k in ("no", "nd"):
        assert torch.equal(o[k], fp32_out[k]) and torch.equal(o16[k], o[k].to(torch.bfloat16))
    np.testing.assert_allclose(o["no"].cpu().numpy(), (obs[idx] - om) / os_, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(o["na"].cpu().numpy()[:, :29], (act[idx][:, :29] - am) / as_, rtol=1e-6, atol=1e-6)
    assert np.all(o["na"].cpu().numpy()[:, 29:] == 0) and np.all(o["nd"].cpu().numpy()[:, 114:] == 0)
    np.testing.assert_allclose(o["nd"].cpu().numpy()[:, :114], (dd[idx] - do[idx])[:, :114] / np.maximum(ma[:114], 1e-4), rtol=1e-6, atol=1e-6)
    for k, kk in (("lp", "logp"), ("ad", "adv"), ("tv", "tar"), ("mk", "mask")):
        assert np.array_equal(o[k].cpu().numpy(), sc[kk][idx])


@pytest.mark.parametrize("trainable_std", [False, True])
def test_actor_loss_head_gradient(trainable_std):
    """(trainable_std: actor_std_type CONSTANT -- per-dimension standard deviations, and d loss / d logstd against autograd.)"""
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(7)
    M = 1000
    mean = (rng.standard_normal((M, 29)) * 0.7).astype(F)
    mean[:20] *= 3  # bound violations
    na = (mean + rng.standard_normal((M, 29)) * 0.05).astype(F)
    adv = np.clip(rng.standard_normal(M), -4, 4).astype(F)
    mask = (rng.rand(M) < 0.9).astype(F)
    c, std = logp_const()
    mt = torch.tensor(mean, requires_grad=True)
    dist, gls = None, torch.zeros(32, device="cuda")
    if trainable_std:
        lst = torch.tensor((np.log(0.05) + rng.uniform(-0.4, 0.4, 29)).astype(F), requires_grad=True)
        na = (mean + rng.standard_normal((M, 29)) * np.exp(lst.detach().numpy())).astype(F)
        d = (torch.tensor(na) - mt) / torch.exp(lst)
        logp = -0.5 * torch.sum(d * d, -1) + (-0.5 * 29 * np.log(2.0 * np.pi) - torch.sum(lst))  # distribution_gaussian_diag.py:90-94
        dist_t = torch.zeros(L.DIST_FLOATS, device="cuda")
        L.call("addhip_dist_refresh", P(T(lst.detach().numpy())), L.ptr(dist_t), L.current_stream())
        dist = L.ptr(dist_t)
    else:
        d = (torch.tensor(na) - mt) / std
        logp = -0.5 * torch.sum(d * d, -1) + c
    old = (logp.detach() + torch.tensor(rng.standard_normal(M).astype(F) * 0.3)).numpy()
    sel = torch.tensor(mask) == 1.0
    ratio = torch.exp(logp - torch.tensor(old))[sel]
    a = torch.tensor(adv)[sel]
    loss = -torch.mean(torch.minimum(a * ratio, a * torch.clamp(ratio, 0.8, 1.2)))
    vmin, vmax = torch.clamp_max(mt[sel] + 1, 0), torch.clamp_min(mt[sel] - 1, 0)
    bound = torch.mean(torch.sum(vmin ** 2, -1) + torch.sum(vmax ** 2, -1))
    reg = torch.mean(torch.sum(mt[sel] ** 2, -1))  # action_reg_weight term (ppo_agent.py:268-272)
    (loss + 10.0 * bound + 0.3 * reg).backward()
    pad = lambda x: np.concatenate([x, np.zeros((M, 3), F)], -1)
    nv = torch.zeros(1, device="cuda")
    dm = torch.zeros(M, 32, device="cuda")
    stats = torch.zeros(8, device="cuda")
    dmask = T(mask)
    L.call("addhip_count_mask", L.ptr(dmask), M, L.ptr(nv), L.current_stream())
    L.call("addhip_actor_loss", P(T(pad(mean))), P(T(pad(na))), P(T(old)), P(T(adv)), L.ptr(dmask), M, std, c, dist, 0.2, 10.0, 0.3, 1.0, L.ptr(nv),
           L.ptr(dm), L.ptr(gls) if trainable_std else None, L.ptr(stats), 32, None, 0.0, L.current_stream())
    torch.cuda.synchronize()
    assert float(nv.item()) == mask.sum()
    if trainable_std:
        gl = lst.grad.numpy()
        np.testing.assert_allclose(gls.cpu().numpy()[:29], gl, rtol=2e-4, atol=2e-4 * np.abs(gl).max())
        assert float(gls[29:].abs().max()) == 0.0
    g = mt.grad.numpy()
    np.testing.assert_allclose(dm.cpu().numpy()[:, :29], g, rtol=2e-4, atol=1e-6 + 2e-4 * np.abs(g).max())
    s = stats.cpu().numpy()
    nvf = mask.sum()  # the statistics are means over the exploring samples already
    np.testing.assert_allclose(-s[0], loss.item(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(s[3], bound.item(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(s[5], reg.item(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(s[2], ratio.mean().item(), rtol=1e-4)
    np.testing.assert_allclose(s[1], (torch.abs(ratio - 1) > 0.2).float().mean().item(), atol=2.0 / nvf)


@pytest.mark.parametrize("ENT_W", [0.0, 0.05])
def test_actor_loss_with_a_log_std_head(ENT_W):
    """(ENT_W: the entropy bonus -w * mean(entropy) of ppo_agent.py:262-266 with per-sample entropies.)  actor_std_type VARIABLE: mean and per-sample log-std as the two halves of a 64-wide head output; d loss / d mean in columns 0..28 and
    d loss / d logstd in columns 32..60 of d_mean against autograd (distribution_gaussian_diag.py:52-53, 90-94; ppo_agent.py:221-275)."""
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(9)
    M = 777
    mean = (rng.standard_normal((M, 29)) * 0.7).astype(F)
    ls = (np.log(0.05) + rng.uniform(-0.5, 0.5, (M, 29))).astype(F)
    na = (mean + rng.standard_normal((M, 29)) * 1.5 * np.exp(ls)).astype(F)
    adv = np.clip(rng.standard_normal(M), -4, 4).astype(F)
    mask = (rng.rand(M) < 0.9).astype(F)
    mt, lt = torch.tensor(mean, requires_grad=True), torch.tensor(ls, requires_grad=True)
    d = (torch.tensor(na) - mt) / torch.exp(lt)
    logp = -0.5 * torch.sum(d * d, -1) + (-0.5 * 29 * np.log(2.0 * np.pi) - torch.sum(lt, -1))
    old = (logp.detach() + torch.tensor(rng.standard_normal(M).astype(F) * 0.3)).numpy()
    sel = torch.tensor(mask) == 1.0
    ratio = torch.exp(logp - torch.tensor(old))[sel]
    a = torch.tensor(adv)[sel]
    loss = -torch.mean(torch.minimum(a * ratio, a * torch.clamp(ratio, 0.8, 1.2)))
    vmin, vmax = torch.clamp_max(mt[sel] + 1, 0), torch.clamp_min(mt[sel] - 1, 0)
    bound = torch.mean(torch.sum(vmin ** 2, -1) + torch.sum(vmax ** 2, -1))
    ent = torch.mean((torch.sum(lt, -1) + 0.5 * 29 * np.log(2.0 * np.pi * np.e))[sel])  # distribution_gaussian_diag.py:96-99
    (loss + 10.0 * bound - ENT_W * ent).backward()
    out64 = np.zeros((M, 64), F)
    out64[:, :29], out64[:, 32:61] = mean, ls
    na32 = np.zeros((M, 32), F)
    na32[:, :29] = na
    head = T(out64)
    nv, dm, stats, dmask = torch.zeros(1, device="cuda"), torch.full((M, 64), 7.0, device="cuda"), torch.zeros(8, device="cuda"), T(mask)
    L.call("addhip_count_mask", L.ptr(dmask), M, L.ptr(nv), L.current_stream())
    L.call("addhip_actor_loss", L.ptr(head), P(T(na32)), P(T(old)), P(T(adv)), L.ptr(dmask), M, float("nan"), float("nan"), None, 0.2, 10.0, 0.0, 1.0, L.ptr(nv),
           L.ptr(dm), None, L.ptr(stats), 64, L.ptr(head) + 4 * 32, ENT_W, L.current_stream())
    torch.cuda.synchronize()
    got = dm.cpu().numpy()
    gm, gl = mt.grad.numpy(), lt.grad.numpy()
    np.testing.assert_allclose(got[:, :29], gm, rtol=2e-4, atol=2e-4 * np.abs(gm).max())
    np.testing.assert_allclose(got[:, 32:61], gl, rtol=2e-4, atol=2e-4 * np.abs(gl).max())
    assert np.all(got[:, 29:32] == 0) and np.all(got[:, 61:] == 0)
    np.testing.assert_allclose(-stats.cpu().numpy()[0], loss.item(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(stats.cpu().numpy()[6], ent.item() if ENT_W else 0.0, rtol=1e-5)


def test_critic_and_disc_heads_and_grad_penalty():
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(8)
    M, K = 999, 512
    H = np.maximum(rng.standard_normal((M + 1, K)), 0).astype(F)
    w, b = (rng.standard_normal(K) * 0.1).astype(F), np.asarray([0.3], F)
    tar = rng.standard_normal(M).astype(F)
    dH, dw, db = T(H), T(w), T(b)
    # critic (ppo_agent.py:209-219)
    dZ, dv, st = torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda"), torch.zeros(8, device="cuda")
    L.call("addhip_critic_head", L.ptr(dH), K, K, M, L.ptr(dw), L.ptr(db), P(T(tar)), 1.0, L.ptr(dZ), L.ptr(dv), L.ptr(st), L.current_stream())
    Ht = torch.tensor(H[:M], requires_grad=True)
    wt, bt = torch.tensor(w, requires_grad=True), torch.tensor(b, requires_grad=True)
    v = Ht @ wt + bt
    loss = torch.mean((torch.tensor(tar) - v) ** 2)
    loss.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(st.cpu().numpy()[0] / M, loss.item(), rtol=1e-5)
    np.testing.assert_allclose(dZ.cpu().numpy(), Ht.grad.numpy() * (H[:M] > 0), rtol=1e-4, atol=1e-8)
    dwv = torch.zeros(K, device="cuda")
    L.call("addhip_weighted_col_sum", L.ptr(dv), L.ptr(dH), K, K, M, L.ptr(dwv), 1.0, 0, L.current_stream())
    np.testing.assert_allclose(dwv.cpu().numpy(), wt.grad.numpy(), rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(dv.cpu().numpy().sum(), bt.grad.item(), rtol=1e-3, atol=1e-6)
    # discriminator head (add_agent.py:141-163, amp_agent.py:177-192); row M is the zero-difference sample
    dl, st2 = torch.zeros(M + 1, device="cuda"), torch.zeros(8, device="cuda")
    L.call("addhip_disc_head", L.ptr(dH), K, K, M, L.ptr(dH) + M * K * 4, L.ptr(dw), L.ptr(db), 0.5, L.ptr(dl), L.ptr(dl) + M * 4, L.ptr(st2), L.current_stream())
    Ht = torch.tensor(H, requires_grad=True)
    logit = Ht @ torch.tensor(w) + torch.tensor(b)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    lneg, lpos = bce(logit[:M], torch.full((M,), 0.1)), bce(logit[M:], torch.full((1,), 0.9))
    (0.5 * 0.5 * (lneg + lpos)).backward()
    torch.cuda.synchronize()
    s = st2.cpu().numpy()
    np.testing.assert_allclose([s[0] / M, s[1]], [lneg.item(), lpos.item()], rtol=1e-5)
    np.testing.assert_allclose(s[2] / M, logit[:M].mean().item(), rtol=1e-4, atol=1e-6)
    assert s[4] == float((logit[:M] < 0).sum()) and s[5] == float(logit[M] > 0)
    out = torch.zeros(M + 1, K, device="cuda")
    L.call("addhip_outer_mask", L.ptr(dl), L.ptr(dw), L.ptr(dH), K, K, M + 1, L.ptr(out), L.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), Ht.grad.numpy() * (H > 0), rtol=1e-4, atol=1e-9)
    # the fused backward of a scalar head: same dZ, plus the head-weight / head-bias / layer-bias gradients, accumulated
    out2 = torch.full((M + 1, K), 7.0, device="cuda")
    gW, gb, gt_ = torch.full((K,), 0.25, device="cuda"), torch.full((1,), 0.25, device="cuda"), torch.full((K,), 0.25, device="cuda")
    out16 = torch.full((M + 1, K), 7.0, device="cuda", dtype=torch.bfloat16)
    L.call("addhip_head_backward", L.ptr(dl), L.ptr(dw), L.ptr(dH), K, K, M + 1, L.ptr(out2), L.ptr(out16), L.STORE_BF16, L.ptr(gW), L.ptr(gb), L.ptr(gt_), None, None, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(out2, out) and torch.equal(out16, out.to(torch.bfloat16))  # the optional bf16 copy: the same values, rounded to nearest even
    dl64, H64, o64 = dl.cpu().numpy().astype(np.float64), H.astype(np.float64), out.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(gW.cpu().numpy() - 0.25, dl64 @ H64, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gb.cpu().numpy() - 0.25, dl64.sum(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gt_.cpu().numpy() - 0.25, o64.sum(0), rtol=1e-4, atol=1e-6)
    L.call("addhip_head_backward", L.ptr(dl), L.ptr(dw), L.ptr(dH), K, K, M + 1, None, None, 0, None, None, None, None, None, L.current_stream())  # every output optional
    # a2 of the gradient-penalty chain: fp32 and / or bf16
    a2, a2_16 = torch.zeros(M, K, device="cuda"), torch.zeros(M, K, device="cuda", dtype=torch.bfloat16)
    L.call("addhip_bcast_mask", L.ptr(dw), L.ptr(dH), K, K, M, L.ptr(a2), L.ptr(a2_16), L.STORE_BF16, None, L.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(a2.cpu().numpy(), np.where(H[:M] > 0, dw.cpu().numpy()[None, :], 0).astype(F))
    assert torch.equal(a2_16, a2.to(torch.bfloat16))
    a2_16b = torch.zeros_like(a2_16)
    L.call("addhip_bcast_mask", L.ptr(dw), L.ptr(dH), K, K, M, None, L.ptr(a2_16b), L.STORE_BF16, None, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(a2_16b, a2_16)
    # critic head without its dZ pass (dZ = NULL)
    dv2, st4 = torch.zeros(M, device="cuda"), torch.zeros(4, device="cuda")
    L.call("addhip_critic_head", L.ptr(dH), K, K, M, L.ptr(dw), L.ptr(db), P(T(tar)), 1.0, None, L.ptr(dv2), L.ptr(st4), L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(dv2, dv)
    np.testing.assert_allclose(st4.cpu().numpy()[0], st.cpu().numpy()[0], rtol=1e-5)  # atomics: summation order differs
    # gradient penalty (add_agent.py:166-178)
    g = np.zeros((M, 116), F)
    g[:, :114] = rng.standard_normal((M, 114)) * 0.1
    g[0, :114] = 0  # |g| = 0 row: sqrt(eps) branch
    gt = torch.tensor(g[:, :114], requires_grad=True)
    n = torch.sqrt(torch.sum(gt * gt, -1) + 1e-8)
    gp = torch.mean((n - 1) ** 2)
    (20.0 * 0.5 * gp).backward()
    G, st3 = torch.ones(M, 116, device="cuda"), torch.zeros(8, device="cuda")
    G16 = torch.ones(M, 116, device="cuda", dtype=torch.bfloat16)
    L.call("addhip_grad_penalty", P(T(g)), 116, 114, M, 10.0, L.ptr(G), L.ptr(G16), L.STORE_BF16, L.ptr(st3), None, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(G16, G.to(torch.bfloat16))
    np.testing.assert_allclose(st3.cpu().numpy()[0] / M, gp.item(), rtol=1e-5)
    np.testing.assert_allclose(G.cpu().numpy()[:, :114], gt.grad.numpy(), rtol=1e-4, atol=1e-8)
    assert np.all(G.cpu().numpy()[:, 114:] == 0)


def test_adamw_matches_torch():
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(9)
    n = 100003
    p0 = rng.standard_normal(n).astype(F)
    p = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.AdamW([p], 1e-4, weight_decay=0.0)
    dp, m, v = T(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = (rng.standard_normal(n) * (10.0 ** rng.randint(-4, 2, n))).astype(F)
        p.grad = torch.tensor(g)
        opt.step()
        L.call("addhip_adamw", L.ptr(dp), P(T(g)), L.ptr(m), L.ptr(v), n, 1e-4, 0.9, 0.999, 1e-8, 0.0, step, L.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(dp.cpu().numpy(), p.detach().numpy(), rtol=1e-6, atol=1e-8)


def test_sgd_momentum_matches_torch():
    """optimizer.type "SGD" (mp_optimizer.py:33-36): torch.optim.SGD(lr, momentum=0.9, weight_decay) over three steps."""
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(2)
    n = 100003
    p0 = rng.standard_normal(n).astype(F)
    pt = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.SGD([pt], 1e-2, momentum=0.9, weight_decay=1e-3)
    dp, buf = T(p0), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(F)
        pt.grad = torch.tensor(g)
        opt.step()
        L.call("addhip_sgd", L.ptr(dp), P(T(g)), L.ptr(buf), n, 1e-2, 0.9, 1e-3, step, L.current_stream())
    torch.cuda.synchronize()
    # (fused vs separate multiply-add roundings: a few ulp of the operands, visible as relative error only where terms cancel)
    np.testing.assert_allclose(dp.cpu().numpy(), pt.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(buf.cpu().numpy(), opt.state[pt]["momentum_buffer"].numpy(), rtol=1e-6, atol=1e-6)


def test_philox_fills_and_return_tracker():
    import torch
    import add_gym_amd._lib as L

    n = 1 << 20
    a, b, u = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    L.call("addhip_fill_normal", L.ptr(a), n, 123, 7, L.current_stream())
    L.call("addhip_fill_normal", L.ptr(b), n, 123, 7, L.current_stream())
    L.call("addhip_fill_uniform", L.ptr(u), n, 123, 8, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(a, b)  # counter-based: same (seed, stream) -> same numbers
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1) < 5e-3 and abs((a ** 4).mean().item() - 3) < 0.05
    assert 0 <= u.min().item() and u.max().item() < 1 and abs(u.mean().item() - 0.5) < 2e-3
    ep = np.zeros((5, 3), F)
    ep[1] = [6.0, 30.0, 3.0]
    ep[3] = [2.0, 8.0, 1.0]
    state = T(np.asarray([2.0, 1.5, 12.0], F))
    L.call("addhip_return_tracker_fold", P(T(ep)), 5, L.ptr(state), L.current_stream())
    torch.cuda.synchronize()
    m1 = (3 / 5) * 2.0 + (2 / 5) * 1.5
    m2 = (1 / 6) * 2.0 + (5 / 6) * m1
    np.testing.assert_allclose(state.cpu().numpy(), [6.0, m2, (1 / 6) * 8 + (5 / 6) * ((3 / 5) * 10 + (2 / 5) * 12)], rtol=1e-6)


def test_grad_clip_matches_torch_clip_grad_norm():
    """MPOptimizer._clip_grads (mp_optimizer.py:45-46)."""
    import torch
    import add_gym_amd._lib as L

    rng = np.random.default_rng(11)
    g = rng.standard_normal(100003).astype(F) * 0.01
    for max_norm in (0.5, 100.0):  # clipping / not clipping
        t = torch.tensor(g, requires_grad=True)
        t.grad = torch.tensor(g)
        want_norm = float(torch.nn.utils.clip_grad_norm_([t], max_norm))
        d = T(g)
        scratch, norm = torch.zeros(4, device="cuda"), torch.zeros(1, device="cuda")
        L.call("addhip_grad_clip", L.ptr(d), g.size, max_norm, L.ptr(scratch), L.ptr(norm), L.current_stream())
        torch.cuda.synchronize()
        np.testing.assert_allclose(norm.item(), want_norm, rtol=1e-6)
        np.testing.assert_allclose(d.cpu().numpy(), t.grad.numpy(), rtol=2e-6, atol=0)


def test_full_size_td_lambda_and_flat_optimiser():
    """BASELINE configs[1] sizes (T=32, N=4096 envs; 4.35 M parameters): TD(lambda) against the oracle recurrence on random
    rewards / values / done flags, and the flat AdamW against torch.optim.AdamW over three steps."""
    import torch
    import add_gym_amd._lib as L
    from oracle import learn as OL

    rng = np.random.default_rng(17)
    Tn, n = 32, 4096
    r = rng.standard_normal((Tn, n)).astype(F)
    vals_all = rng.standard_normal((Tn + 1, n)).astype(F)
    done = rng.choice([0, 0, 0, 0, 0, 0, 1, 2], size=(Tn, n)).astype(np.int32)
    mask = (rng.random((Tn, n)) < 0.9).astype(F)
    nxt = vals_all[1:].copy()
    nxt[(done == 1) | (done == 2)] = 0.0
    want = OL.td_lambda_return(r, nxt, done, 0.99, 0.95)
    va = T(vals_all)
    tar, adv = torch.zeros(Tn, n, device="cuda"), torch.zeros(Tn, n, device="cuda")
    scratch, stats = torch.zeros(4096, dtype=torch.float64, device="cuda"), torch.zeros(2, device="cuda")
    L.call("addhip_td_lambda_adv", P(T(r)), L.ptr(va[1:]), None, L.ptr(va), P(T(done, torch.int32)), P(T(mask)), Tn, n, 0.99, 0.95, 0.0, 0.0, 4.0,
           L.ptr(tar), L.ptr(adv), L.ptr(scratch), L.ptr(stats), L.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(tar.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    a = (want - vals_all[:-1])[mask == 1]
    np.testing.assert_allclose(stats.cpu().numpy(), [a.mean(dtype=np.float64), a.std(ddof=1, dtype=np.float64)], rtol=1e-4, atol=1e-6)
    norm = np.clip((want - vals_all[:-1] - stats[0].item()) / max(stats[1].item(), 1e-5), -4.0, 4.0)
    np.testing.assert_allclose(adv.cpu().numpy(), norm, rtol=1e-4, atol=1e-4)

    count = 4349983 + 1361  # parameters + layout padding
    p0 = (rng.standard_normal(count) * 0.05).astype(F)
    p = torch.nn.Parameter(torch.tensor(p0))
    opt = torch.optim.AdamW([p], 1e-4, weight_decay=0.0)
    dp, m, v = T(p0), torch.zeros(count, device="cuda"), torch.zeros(count, device="cuda")
    for step in range(1, 4):
        g = (rng.standard_normal(count) * 1e-3).astype(F)
        p.grad = torch.tensor(g)
        opt.step()
        L.call("addhip_adamw", L.ptr(dp), P(T(g)), L.ptr(m), L.ptr(v), count, 1e-4, 0.9, 0.999, 1e-8, 0.0, step, L.current_stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(dp.cpu().numpy(), p.detach().numpy(), rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("kind", ["adamw", "sgd"])
def test_optimizer_step_is_the_separate_kernels_in_one_launch(kind):
    """addhip_optimizer_step (MPOptimizer.step, mp_optimizer.py:14-46, as one launch) == addhip_adamw / addhip_sgd element for element, plus the
    bf16 shadow of the new parameters (== addhip_to_bf16 of them) and the zeroed gradient; a count that is not a multiple of 4."""
    import torch
    import add_gym_amd._lib as L

    n = 4 * 50001 + 3
    rng = np.random.RandomState(2)
    p0 = torch.tensor(rng.standard_normal(n + 1).astype(F), device="cuda")[:n]
    bufs = [[p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")] for _ in range(2)]
    p16 = torch.zeros(n + 1, dtype=torch.bfloat16, device="cuda")
    for step in range(1, 4):
        g = torch.tensor((rng.standard_normal(n) * 0.1).astype(F), device="cuda")
        (pa, ma, va), (pb, mb, vb) = bufs
        if kind == "adamw":
            L.call("addhip_adamw", L.ptr(pa), L.ptr(g), L.ptr(ma), L.ptr(va), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, step, L.current_stream())
        else:
            L.call("addhip_sgd", L.ptr(pa), L.ptr(g), L.ptr(ma), n, 1e-2, 0.9, 1e-3, step, L.current_stream())
        g2 = g.clone()
        o = L.OptimizerT(L.OPT_ADAMW if kind == "adamw" else L.OPT_SGD, L.ptr(pb), L.ptr(g2), L.ptr(mb), L.ptr(vb) if kind == "adamw" else None, n,
                         1e-3 if kind == "adamw" else 1e-2, 0.9, 0.999, 1e-8, 1e-2 if kind == "adamw" else 1e-3, step, L.ptr(p16), 1)
        L.call("addhip_optimizer_step", o, L.current_stream())
        torch.cuda.synchronize()
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), step
        assert float(g2.abs().max()) == 0.0
        assert torch.equal(p16[:n], pb.to(torch.bfloat16)) and float(p16[n]) == 0.0
    # without the options the gradient is left alone
    g3 = g.clone()
    o.grad, o.param16, o.zero_grad, o.step = L.ptr(g3), None, 0, 4
    L.call("addhip_optimizer_step", o, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(g3, g)


def test_plane_storage_outputs_of_the_producers():
    """Plane storage (include/addhip.h: ADDHIP_STORE_BF16X3) written by the kernels that produce GEMM operands in the update step --
    gather, head backward, a2 of the penalty chain, the penalty gradient, the optimiser's parameter shadow: each is the exact three-way
    split (tests/util.py: to_planes) of the fp32 value the same launch writes."""
    import torch
    import add_gym_amd._lib as L
    from util import to_planes

    X3 = L.STORE_BF16X3
    rng = np.random.RandomState(16)
    i16 = lambda r, c: torch.full((r, 3 * c), 5, dtype=torch.int16, device="cuda")
    pl = lambda t: t.cpu().numpy().view(np.uint16)
    # gather: obs rows of stride 272 (264 used), difference rows of stride 128 (114 used)
    R, Mb = 2000, 555
    obs, do, dd = np.zeros((R, 272), F), np.zeros((R, 128), F), np.zeros((R, 128), F)
    obs[:, :264], do[:, :114], dd[:, :114] = rng.standard_normal((R, 264)), rng.standard_normal((R, 114)), rng.standard_normal((R, 114))
    act = np.zeros((R, 32), F)
    sc = {k: rng.standard_normal(R).astype(F) for k in ("logp", "adv", "tar", "mask")}
    om, os_ = np.zeros(272, F), np.ones(272, F)
    om[:264], os_[:264] = rng.standard_normal(264), rng.rand(264) + 0.5
    am, as_ = np.zeros(32, F), np.ones(32, F)
    ma = (rng.rand(128) * 0.3 + 0.01).astype(F)
    idx = rng.randint(0, R, Mb).astype(np.int64)
    d = {k: T(v) for k, v in dict(obs=obs, act=act, do=do, dd=dd, om=om, os=os_, am=am, as_=as_, ma=ma, idx=idx, **sc).items()}
    z = lambda *s: torch.zeros(*s, device="cuda")
    o = dict(no=z(Mb, 272), na=z(Mb, 32), lp=z(Mb), ad=z(Mb), tv=z(Mb), mk=z(Mb), nd=z(Mb, 128), no16=i16(Mb, 272), nd16=i16(Mb, 128))
    g = L.GatherT(L.ptr(d["idx"]), Mb, L.ptr(d["obs"]), 272, 264, L.ptr(d["om"]), L.ptr(d["os"]), L.ptr(d["act"]), L.ptr(d["am"]), L.ptr(d["as_"]),
                  L.ptr(d["logp"]), L.ptr(d["adv"]), L.ptr(d["tar"]), L.ptr(d["mask"]), L.ptr(d["do"]), L.ptr(d["dd"]), 128, 114, L.ptr(d["ma"]), 1e-4,
                  L.ptr(o["no"]), L.ptr(o["na"]), L.ptr(o["lp"]), L.ptr(o["ad"]), L.ptr(o["tv"]), L.ptr(o["mk"]), L.ptr(o["nd"]), L.ptr(o["no16"]), L.ptr(o["nd16"]), X3)
    am = torch.zeros(5, L.AMAX_SLOTS, dtype=torch.int32, device="cuda")  # tracked maxima (ADDHIP_PREC_F16X2): obs, diff, dZ, a2, G
    amax = lambda i: float(am[i].cpu().numpy().view(np.float32).max())
    g.obs_amax, g.diff_amax = L.ptr(am[0]), L.ptr(am[1])
    L.call("addhip_gather_minibatch", g, L.current_stream())
    torch.cuda.synchronize()
    assert amax(0) == float(o["no"].abs().max()) and amax(1) == float(o["nd"].abs().max())
    assert np.array_equal(pl(o["no16"]), to_planes(o["no"].cpu().numpy())) and np.array_equal(pl(o["nd16"]), to_planes(o["nd"].cpu().numpy()))
    assert float(o["no"].abs().max()) > 1.0 and float(o["nd"].abs().max()) > 1.0
    # head backward, a2, penalty gradient
    M, K = 999, 512
    H = T(np.maximum(rng.standard_normal((M, K)), 0).astype(F))
    w, v = T((rng.standard_normal(K) * 0.1).astype(F)), T((rng.standard_normal(M) * 1e-3).astype(F))
    dZ, dZ16 = z(M, K), i16(M, K)
    L.call("addhip_head_backward", L.ptr(v), L.ptr(w), L.ptr(H), K, K, M, L.ptr(dZ), L.ptr(dZ16), X3, None, None, None, L.ptr(am[2]), None, L.current_stream())
    a2, a2_16 = z(M, K), i16(M, K)
    L.call("addhip_bcast_mask", L.ptr(w), L.ptr(H), K, K, M, L.ptr(a2), L.ptr(a2_16), X3, L.ptr(am[3]), L.current_stream())
    gsrc = np.zeros((M, 128), F)
    gsrc[:, :114] = rng.standard_normal((M, 114)) * 0.1
    G, G16, st = z(M, 128), i16(M, 128), z(8)
    L.call("addhip_grad_penalty", P(T(gsrc)), 128, 114, M, 10.0, L.ptr(G), L.ptr(G16), X3, L.ptr(st), L.ptr(am[4]), L.current_stream())
    torch.cuda.synchronize()
    for i, (fp, p16) in enumerate(((dZ, dZ16), (a2, a2_16), (G, G16))):
        assert np.array_equal(pl(p16), to_planes(fp.cpu().numpy())) and float(fp.abs().max()) > 0.0
        assert amax(2 + i) == float(fp.abs().max())
    with pytest.raises(RuntimeError):  # plane storage needs rows of whole 8-value groups
        L.call("addhip_grad_penalty", P(T(gsrc[:, :116].copy())), 116, 114, M, 10.0, L.ptr(G), L.ptr(G16), X3, L.ptr(st), None, L.current_stream())
    # the optimiser's shadow of the new parameters
    n = 8 * 30001
    p = T(rng.standard_normal(n).astype(F))
    gr, m1, m2 = T((rng.standard_normal(n) * 0.1).astype(F)), z(n), z(n)
    p16 = torch.full((3 * n,), 5, dtype=torch.int16, device="cuda")
    oc = L.OptimizerT(L.OPT_ADAMW, L.ptr(p), L.ptr(gr), L.ptr(m1), L.ptr(m2), n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1, L.ptr(p16), 1, X3)
    L.call("addhip_optimizer_step", oc, L.current_stream())
    torch.cuda.synchronize()
    assert np.array_equal(pl(p16), to_planes(p.cpu().numpy().reshape(1, -1)).reshape(-1)) and float(gr.abs().max()) == 0.0


def test_fixed_order_reductions_repeat_bit_for_bit():
    """addhip_col_sum_ordered and addhip_head_backward with its ordered scratch (agent.deterministic): the same sums as the atomics forms to
    fp32 accuracy, and bit-identical from launch to launch."""
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(21)
    M, K = 16385, 512
    X = T((rng.standard_normal((M, K)) * 10.0 ** rng.uniform(-3, 3, (M, 1))).astype(F))
    scratch = torch.zeros(L.HEAD_BWD_BLOCKS * (2 * K + 4), device="cuda")
    outs = []
    for _ in range(3):
        o = torch.full((K,), 0.5, device="cuda")
        L.call("addhip_col_sum_ordered", L.ptr(X), M, K, K, L.ptr(o), 2.0, 1, L.ptr(scratch), L.current_stream())
        outs.append(o)
    ref = torch.full((K,), 0.5, device="cuda")
    L.call("addhip_col_sum", L.ptr(X), M, K, K, L.ptr(ref), 2.0, 1, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    x64 = X.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(outs[0].cpu().numpy(), 0.5 + 2.0 * x64.sum(0), rtol=1e-5, atol=1e-3 * np.abs(x64).sum(0).max() * 1e-4)
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-2)
    H = T(np.maximum(rng.standard_normal((M, K)), 0).astype(F))
    w, v = T((rng.standard_normal(K) * 0.1).astype(F)), T(rng.standard_normal(M).astype(F))
    res = []
    for ordered in (scratch, scratch, None):
        gW, gb, gt_ = torch.full((K,), 0.25, device="cuda"), torch.full((1,), 0.25, device="cuda"), torch.full((K,), 0.25, device="cuda")
        dZ = torch.zeros(M, K, device="cuda")
        L.call("addhip_head_backward", L.ptr(v), L.ptr(w), L.ptr(H), K, K, M, L.ptr(dZ), None, 0, L.ptr(gW), L.ptr(gb), L.ptr(gt_), None,
               None if ordered is None else L.ptr(ordered), L.current_stream())
        res.append((gW, gb, gt_, dZ))
    torch.cuda.synchronize()
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    for a, b in zip(res[0], res[2]):  # the atomics form: the same sums in another order
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("hidden,rows,trainable_std", [(512, 4000, False), (256, 129, False), (128, 32, False), (512, 1500, True), (128, 70, True)])
def test_fused_actor_head_equals_the_unfused_sequence(hidden, rows, trainable_std):
    """addhip_actor_head (csrc/actor_head.hip) against the launches it replaces -- head GEMM, addhip_actor_loss, weight-gradient GEMM + slab
    reduce, column sums, dz GEMM with the ReLU mask -- on the same inputs: dz, d loss / d Wh, d bh, the top layer's bias gradient and the
    loss diagnostics; ragged row counts, masked-out samples, action-bound violations."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    rng = np.random.RandomState(31 + hidden)
    M, K = rows, hidden
    H = T(np.maximum(rng.standard_normal((M, K)), 0).astype(F))
    Wh = np.zeros((32, K), F)
    Wh[:29] = rng.standard_normal((29, K)) * 0.05
    bh = np.zeros(32, F)
    bh[:29] = rng.standard_normal(29) * 0.3
    na = np.zeros((M, 32), F)
    na[:, :29] = rng.standard_normal((M, 29))
    old_logp = (rng.standard_normal(M) * 0.3 - 30).astype(F)
    adv = rng.standard_normal(M).astype(F)
    mask = (rng.rand(M) < 0.9).astype(F)
    dWh, dbh, dna, dol, dadv, dmask = T(Wh), T(bh), T(na), T(old_logp), T(adv), T(mask)
    z = lambda *s: torch.zeros(*s, device="cuda")
    st = L.current_stream()
    nv = z(1)
    L.call("addhip_count_mask", L.ptr(dmask), M, L.ptr(nv), st)
    std, logp_const, clip, bw, rw, ls = 0.05, float(-0.5 * 29 * np.log(2 * np.pi) - 29 * np.log(0.05)), 0.2, 10.0, 0.01, 0.5
    dist, want_gls = None, z(32)
    if trainable_std:  # actor_std_type CONSTANT: standard deviations ~1 so that the unit-normal actions above stay a few sigma out
        dist_t = z(L.DIST_FLOATS)
        L.call("addhip_dist_refresh", P(T(rng.uniform(-0.4, 0.4, 29).astype(F))), L.ptr(dist_t), st)
        dist = L.ptr(dist_t)
    # --- the unfused sequence
    mean, d_mean, stats_a = z(M, 32), z(M, 32), z(8)
    L.call("addhip_gemm_f32", gemm(M, 32, K, L.ptr(H), K, 1, L.ptr(dWh), K, 1, L.ptr(mean), 32, L.EPI_BIAS, L.ptr(dbh)), st)
    L.call("addhip_actor_loss", L.ptr(mean), L.ptr(dna), L.ptr(dol), L.ptr(dadv), L.ptr(dmask), M, std, logp_const, dist, clip, bw, rw, ls, L.ptr(nv), L.ptr(d_mean),
           L.ptr(want_gls) if trainable_std else None, L.ptr(stats_a), 32, None, 0.0, st)
    torch.cuda.synchronize()
    dm64, H64 = d_mean.cpu().numpy().astype(np.float64), H.cpu().numpy().astype(np.float64)
    want_gW, want_gb = dm64.T @ H64, dm64.sum(0)
    want_dz = (dm64 @ Wh.astype(np.float64)) * (H64 > 0)
    # --- one launch
    ns = L.load().addhip_actor_head_slabs(M)
    dz, dz16 = z(M, K), torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    SL = L.actor_head_slab(K)
    slabs, gb_top, stats_b = torch.full((ns, SL), 7.0, device="cuda"), z(4, K), z(8)
    amax = torch.zeros(L.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    h = L.ActorHeadT(M, K, L.ptr(H), L.ptr(dWh), L.ptr(dbh), L.ptr(dna), L.ptr(dol), L.ptr(dadv), L.ptr(dmask), L.ptr(nv), std, logp_const, clip, bw, rw, ls, dist,
                     L.ptr(dz), L.ptr(dz16), L.STORE_BF16, L.ptr(slabs), ns, L.ptr(gb_top), 4, K, L.ptr(stats_b), L.ptr(amax))
    L.call("addhip_actor_head", h, st)
    out = z(SL)
    L.call("addhip_slab_reduce", L.ptr(slabs), ns, SL, L.ptr(out), SL, 1.0, 0, st)
    torch.cuda.synchronize()
    # d loss / d logstd: the slab's last 32 floats (zeros without a trainable log-std)
    gl = want_gls.cpu().numpy()
    np.testing.assert_allclose(out[32 * K + 32:].cpu().numpy(), gl, rtol=1e-4, atol=2e-5 * np.abs(gl).max())
    assert trainable_std == (np.abs(gl).max() > 0)
    scale = np.abs(want_dz).max()
    assert scale > 0
    np.testing.assert_allclose(dz.cpu().numpy(), want_dz, rtol=1e-4, atol=2e-5 * scale)
    assert torch.equal(dz16, dz.to(torch.bfloat16))
    assert float(amax.cpu().numpy().view(np.float32).max()) == float(dz.abs().max())
    gW = out[:32 * K].view(32, K).cpu().numpy()
    np.testing.assert_allclose(gW, want_gW, rtol=1e-4, atol=2e-5 * np.abs(want_gW).max())
    np.testing.assert_allclose(out[32 * K:32 * K + 32].cpu().numpy(), want_gb, rtol=1e-4, atol=2e-5 * np.abs(want_gb).max())
    np.testing.assert_allclose(gb_top.sum(0).cpu().numpy(), want_dz.sum(0), rtol=1e-3, atol=1e-4 * np.abs(want_dz).sum(0).max())
    np.testing.assert_allclose(stats_b.cpu().numpy(), stats_a.cpu().numpy(), rtol=2e-5, atol=1e-6)
    assert float(gW[29:].max()) == 0.0 and float(np.abs(gW[29:]).max()) == 0.0  # padding rows of the head receive no gradient

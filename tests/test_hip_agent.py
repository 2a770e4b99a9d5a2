"""Agent-level GPU parity: one minibatch loss/gradient/AdamW step and one complete training iteration
(rollout + build-train-data + update) of the HIP engine against the reference's golden vectors and the CPU oracle,
with the reference's random draws replayed."""
import numpy as np
import pytest

from oracle import learn as OL
from tests.util import DEFAULT_TASK, gload, kin_meta
from tests.test_oracle_vs_golden import _check_summary

pytestmark = pytest.mark.gpu
F = np.float32


def make_cfg(num_envs, steps_per_iter=32, **agent_over):
    import add_gym_amd  # noqa: F401
    from add_gym_amd.config import load_config

    cfg = load_config("train", [f"engine.num_envs={num_envs}", f"agent.steps_per_iter={steps_per_iter}"])
    cfg["agent"].update(agent_over)
    cfg["task"]["motion_joint_order"] = kin_meta()["motion_joint_order"]
    return cfg


def make_agent(cfg, frames_list, weights):
    """ADDAgent on the golden clip frames (MotionLib accepts in-memory frames)."""
    import add_gym_amd.learning.add_agent as A
    from add_gym_amd.anim.motion_lib import MotionLib

    orig = A.MotionLib

    def lib(motion_file, order, kin, dt, dev, reference_compat=True, cache_dir=None):
        return MotionLib(None, order, kin, dt, dev, reference_compat=reference_compat, frames_list=frames_list, weights=weights)

    A.MotionLib = lib
    try:
        return A.ADDAgent(cfg)
    finally:
        A.MotionLib = orig


def T(x, dtype=None):
    import torch

    return torch.tensor(np.ascontiguousarray(x), dtype=dtype, device="cuda")


# agent.matmul_precision product modes with fp32-class error bounds (include/addhip.h); "bf16x3" splits fp32 operands inside every GEMM
# (csrc/gemm_split.hip), "bf16x3_planes" runs the update step on plane-stored operands (csrc/gemm_x3.hip), "f16x2" splits them two ways
# into fp16 on per-tensor power-of-two scales (four products, 22-bit operands: ADDHIP_PREC_F16X2)
PRECISIONS = ["fp32", "bf16x3", "bf16x3_planes", "f16x2", "bf16x2"]


def sync_minibatch16(ag):
    """The 16-bit copies of the gathered minibatch rows that addhip_gather_minibatch writes in the storage modes (bf16: rounded to nearest
    even; bf16x3: plane storage), refreshed from the fp32 rows a test wrote by hand."""
    import add_gym_amd._lib as L

    W = ag._W
    if getattr(ag, "_f16x2", False):  # f16x2: the rows' tracked maxima instead (the fp16 split's scales)
        L.call("addhip_fill_zero", L.ptr(W["mb_amax"]), 2 * L.AMAX_SLOTS, L.current_stream())
        for i, src in enumerate(("norm_obs", "norm_diff")):
            L.call("addhip_amax_f32", L.ptr(W[src]), W[src].numel(), L.ptr(W["mb_amax"][i]), L.current_stream())
    if "norm_obs16" not in W:
        return
    x3 = ag._storage16 == L.STORE_BF16X3
    for src, dst in (("norm_obs", "norm_obs16"), ("norm_diff", "norm_diff16")):
        rows, cols = W[dst].shape[0], W[src].shape[1]
        L.call("addhip_to_bf16x3" if x3 else "addhip_to_bf16", L.ptr(W[src]), L.ptr(W[dst]), rows, cols, cols, cols, L.current_stream())


@pytest.mark.parametrize("precision", PRECISIONS)
def test_minibatch_loss_gradients_and_adamw_match_reference(precision):
    minibatch_loss_gradients_and_adamw(precision)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x2"])  # (fp32-class modes: this fixture's 12-sigma actions amplify bf16 storage rounding to tens of percent)
def test_other_net_modules_match_reference(precision):
    """agent.model.{actor,critic,disc}_net other than add_g1.yaml's (the reference's net registry, nets/net_builder.py:5-11): a 256/128
    actor, a 512/256 critic, a 128/64 discriminator through the same update plan (the 128-wide actor head takes the fused kernel),
    against gradients and parameters the reference produced with those modules (tools/gen_golden_agent.py: gen_losses_small_nets)."""
    minibatch_loss_gradients_and_adamw(precision, "losses_small_nets")


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x2"])  # (fp32-class modes: this fixture's 12-sigma actions amplify bf16 storage rounding to tens of percent)
def test_trainable_log_std_matches_reference(precision):
    """agent.model.actor_std_type CONSTANT (distribution_gaussian_diag.py:32-37): one trainable log-std per action dimension -- its gradient
    through the fused actor head, its AdamW step, the refreshed std / log-probability constant of the next step -- against the reference's
    gradients and parameters (tools/gen_golden_agent.py: gen_losses_constant_std)."""
    minibatch_loss_gradients_and_adamw(precision, "losses_constant_std")


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x2", "bf16x3_planes"])
def test_three_layer_discriminator_matches_reference(precision):
    """agent.model.disc_net = fc_3layers_1024units: the gradient-penalty chain (the hand-derived double backward of add_agent.py:166-178) runs
    through three hidden layers -- a[2] -> a[1] -> a[0] -> g, G, e[0] -> e[1] -> e[2], three extra weight-gradient products -- against the
    reference's gradients and parameters (tools/gen_golden_agent.py: gen_losses_disc3).  bf16x3_planes: the same chain on plane storage."""
    minibatch_loss_gradients_and_adamw(precision, "losses_disc3")


@pytest.mark.parametrize("precision,fixture", [("fp32", "losses_constant_std_entropy"), ("f16x2", "losses_constant_std_entropy"), ("fp32", "losses_variable_std_entropy")])
def test_entropy_bonus_reaches_the_trainable_log_std(precision, fixture):
    """action_entropy_weight = 0.05 (ppo_agent.py:262-266).  actor_std_type CONSTANT: the bonus's gradient, -w on every log-std, is added
    behind the exchange (ADDAgent._post_exchange_grads); VARIABLE: per-sample entropies, -w / n on every log-std of an exploring sample inside
    the loss kernel.  Gradients, three AdamW steps and the logged entropy against the reference (tools/gen_golden_agent.py)."""
    minibatch_loss_gradients_and_adamw(precision, fixture)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x2"])
def test_log_std_head_matches_reference(precision):
    """agent.model.actor_std_type VARIABLE (distribution_gaussian_diag.py:38-43, 52-53): the log-std is a second linear head on the actor's
    last layer -- rows 32..60 of a 64-row head matrix here, so that the two heads are one 64-wide product in the head GEMM, its weight
    gradient and dz -- gradients of both heads, three AdamW steps, against the reference (tools/gen_golden_agent.py: gen_losses_variable_std)."""
    minibatch_loss_gradients_and_adamw(precision, "losses_variable_std")


def test_three_layer_discriminator_under_bf16_storage():
    """The same three-layer chain on bf16 STORAGE (16-bit a[i] / e[i] rows, transposed weight shadows of all three layers): the
    discriminator's gradients stay within bf16 rounding of the oracle's (relative L2 per tensor; this fixture's actor terms are not
    looked at -- its 12-sigma actions amplify 8-bit operands to tens of percent), everything finite."""
    import json
    import torch
    import add_gym_amd._lib as L

    g = gload("losses_disc3")
    nets = json.loads(str(g["nets"]))
    M = g["in.obs"].shape[0]
    cfg = make_cfg(M // 4, steps_per_iter=8, matmul_precision="bf16")
    cfg["agent"]["model"].update(nets)
    ag = make_agent(cfg, [gload("motion_small")["frames"]], [1.0])
    params = OL.synth_params(int(g["seed"]), nets=nets)
    ag._model.load({k: torch.tensor(v) for k, v in params.items()})
    dn = OL.DiffNormalizer(114)
    dn.mean_abs = g["disc_mean_abs"]
    nd = dn.normalize(g["in.disc_obs_demo"] - g["in.disc_obs"])
    W = ag._W
    for k in ("norm_obs", "norm_act", "mb_adv", "mb_tar"):
        W[k].zero_()
    W["mb_logp"].fill_(ag._model.logp_const)
    W["mb_mask"].fill_(1.0)
    W["norm_diff"].zero_()
    W["norm_diff"][:M, :114] = T(nd)
    sync_minibatch16(ag)
    m = ag._model
    W["stats"].zero_()
    m.grads.zero_()
    ag._update_plan.run(L.current_stream())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.grads).all())
    gh = {k: v.numpy() for k, v in m.export(m.grads).items()}
    model = OL.Model(params)
    mb = dict(norm_obs=np.zeros((M, 264), F), norm_action=np.zeros((M, 29), F), a_logp=np.full(M, ag._model.logp_const, F), adv=np.zeros(M, F), tar_val=np.zeros(M, F),
              rand_action_mask=np.ones(M, F), norm_diff=nd)
    loss, _ = OL.compute_loss(model, OL.LossCfg(), mb)
    go = OL.AdamW(model, 1e-4).step(loss)
    for k, v in go.items():
        if "_disc_" in k:
            rel = np.linalg.norm(gh[k] - v) / (np.linalg.norm(v) + 1e-12)
            assert rel <= 8e-2, (k, float(rel))


def minibatch_loss_gradients_and_adamw(precision="fp32", fixture="losses"):
    """(At this fixture's 256 rows every GEMM is below the size from which the bf16-MFMA kernels are dispatched, so the three
    modes run the same kernels here; the modes themselves are pinned at full size in tests/test_hip_fullsize.py.)"""
    import torch
    import add_gym_amd._lib as L

    import json

    g = gload(fixture)
    nets = json.loads(str(g["nets"])) if "nets" in g.files else None
    M = g["in.obs"].shape[0]
    cfg = make_cfg(M // 4, steps_per_iter=8, matmul_precision=precision)
    cfg["agent"]["model"].update(nets or {})
    agent_over = json.loads(str(g["agent_over"])) if "agent_over" in g.files else {}
    cfg["agent"].update(agent_over)
    from tests.test_oracle_vs_golden import golden_logstd

    logstd = golden_logstd(g)
    if logstd:
        cfg["agent"]["model"]["actor_std_type"] = "VARIABLE" if logstd == "variable" else "CONSTANT"
    ag = make_agent(cfg, [gload("motion_small")["frames"]], [1.0])
    assert ag.Mb == M
    params = OL.synth_params(int(g["seed"]), nets=nets, logstd=logstd)
    ag._model.load({k: torch.tensor(v) for k, v in params.items()})
    on = OL.Normalizer(264, g["obs_mean"], g["obs_std"])
    an = OL.Normalizer(29, g["a_mean"], g["a_std"])
    dn = OL.DiffNormalizer(114)
    dn.mean_abs = g["disc_mean_abs"]
    mb = dict(norm_obs=on.normalize(g["in.obs"]), norm_action=an.normalize(g["in.action"]), a_logp=g["in.a_logp"], adv=g["in.adv"],
              tar_val=g["in.tar_val"], rand_action_mask=g["in.rand_action_mask"], norm_diff=dn.normalize(g["in.disc_obs_demo"] - g["in.disc_obs"]))
    W = ag._W
    W["norm_obs"].zero_()
    W["norm_obs"][:, :264] = T(mb["norm_obs"])
    W["norm_act"].zero_()
    W["norm_act"][:, :29] = T(mb["norm_action"])
    W["mb_logp"].copy_(T(mb["a_logp"]))
    W["mb_adv"].copy_(T(mb["adv"]))
    W["mb_tar"].copy_(T(mb["tar_val"]))
    W["mb_mask"].copy_(T(mb["rand_action_mask"]))
    W["norm_diff"].zero_()
    W["norm_diff"][:M, :114] = T(mb["norm_diff"])
    sync_minibatch16(ag)
    model = OL.Model(params)
    opt = OL.AdamW(model, 1e-4)
    st = L.current_stream()
    m = ag._model
    for step in range(3):
        W["stats"].zero_()
        m.grads.zero_()  # zero_grad: _run_update_sections does it before the sections fork
        ag._update_plan.run(st)
        ag._post_exchange_grads(st)  # (what _update_model adds behind the exchange: the entropy bonus of a trainable log-std)
        torch.cuda.synchronize()
        grads_hip = {k: v.numpy() for k, v in m.export(m.grads).items() if k != OL.LOGSTD_KEY or logstd}
        loss, info = OL.compute_loss(model, OL.LossCfg(**agent_over), mb)
        grads_orc = opt.step(loss)
        if step == 0:
            # (1) against the oracle, tensor by tensor, full gradients
            for k, go in grads_orc.items():
                gh = grads_hip[k]
                scale = np.abs(go).max() + 1e-12
                # (the fixture's actions lie ~12 sigma out: |logp| ~ 2e3, where one fp32 ulp is 1.2e-4 -- the importance ratio, and with it
                # every actor gradient, carries that much noise in ANY fp32 evaluation order, torch's included: tests/test_oracle_vs_golden.py)
                # (measured over the precision modes: up to 3.1e-4 of a tensor's largest gradient; the golden summaries below are held to 5e-4)
                tol = 4e-4 if "_actor_layers" in k or "_action_dist" in k else 2e-4
                assert np.abs(gh - go).max() <= tol * scale + 1e-9, (k, float(np.abs(gh - go).max()), float(scale))
            # (2) against the reference's own gradients (golden summaries)
            _check_summary(g, "grad", grads_hip, rtol=5e-4)
            # logged scalars of the step
            ag._total_samples = ag.T * ag.N
            stats = ag._collect_info(1)
            # (this fixture masks 10 % of the samples out of the actor terms: the means run over the exploring samples)
            for k in ("critic_loss", "actor_loss", "disc_loss", "disc_grad_penalty", "disc_logit_loss", "disc_pos_acc", "disc_neg_acc", "disc_pos_logit",
                      "disc_neg_logit", "clip_frac", "imp_ratio", "action_bound_loss", "action_entropy"):
                if "info." + k in g.files:
                    np.testing.assert_allclose(stats[k], float(g["info." + k]), rtol=2e-4, atol=2e-4, err_msg=k)
        m.opt_step += 1
        L.call("addhip_adamw", L.ptr(m.params), L.ptr(m.grads), L.ptr(m.exp_avg), L.ptr(m.exp_avg_sq), m.count, 1e-4, 0.9, 0.999, 1e-8, 0.0, m.opt_step, st)
        m.refresh_shadow()  # (what the agent's own optimiser step keeps current: the plane-storage shadow, the tracked parameter maximum)
        torch.cuda.synchronize()
        if step in (0, 2):
            ph = {k: v.numpy() for k, v in m.export().items() if k != OL.LOGSTD_KEY or logstd}
            # Adam's early steps move every weight by ~lr*sign(g): where a gradient is ~0 its sign is rounding noise, so
            # a few elements may differ by up to 2*lr per step; everything else must agree to fp32 rounding
            _check_param_summary(g, f"param{step + 1}", ph, step + 1)
            for k, v in model.p.items():
                d = np.abs(ph[k] - v.detach().numpy())
                assert d.max() <= 2.1e-4 * (step + 1), (k, float(d.max()))
                assert (d > 5e-6).mean() < 0.01, (k, float((d > 5e-6).mean()))
    # padded rows / columns of the device layout never receive gradient
    wh = m.view("actor", "Wh", m.grads)  # (a log-std head occupies rows 32..60)
    assert float(wh[29:32].abs().max()) == 0 and float(wh[61:].abs().max() if wh.shape[0] == 64 else 0.0) == 0
    assert float(m.view("disc", "W0", m.grads)[:, 114:].abs().max()) == 0


def _check_param_summary(g, prefix, named, steps, lr=1e-4, long_run=False, mean_frac=0.01):
    """Parameters after `steps` Adam steps against the reference's summaries.  Adam's early steps move each weight by
    ~lr*sign(g); where a gradient is ~0 its sign is rounding noise, so individual elements may differ by up to 2*lr per
    step while everything else agrees to fp32 rounding."""
    for name, val in named.items():
        f = np.asarray(val, np.float64).reshape(-1)
        ref_l2 = float(g[f"{prefix}.{name}.l2"])
        # (long runs: + one lr of absolute slack -- a 29-element bias whose gradients are atomically accumulated sums near 0 moves
        # by +-lr per step on rounding noise, which is a visible fraction of its own small norm)
        assert abs(np.sqrt(np.square(f).sum()) - ref_l2) <= (6e-4 if long_run else 2e-4) * max(ref_l2, 1e-12) + (lr if long_run else 0.0), (prefix, name)
        stride = max(1, f.size // 64)
        d = np.abs(np.asarray(val, F).reshape(-1)[::stride][:64] - g[f"{prefix}.{name}.sample"])
        if long_run:  # a whole iteration (40 steps): rounding noise compounds; bound it against the distance travelled (<= lr*steps)
            # (elements whose gradient hovers around 0 random-walk by +-lr per step: allow them the full travel, few of them)
            assert d.max() <= lr * steps and np.percentile(d, 90) <= 0.05 * lr * steps and d.mean() <= mean_frac * lr * steps, \
                (prefix, name, float(d.max()), float(d.mean()))
            continue
        assert d.max() <= 2.1 * lr * steps, (prefix, name, float(d.max()))
        assert (d > 5e-6).mean() <= 0.08, (prefix, name, float((d > 5e-6).mean()))


def _mid(cdf, k):
    lo = 0.0 if k == 0 else float(cdf[k - 1])
    return F(0.5 * (lo + float(cdf[k])))


LOOP_VARIANTS = {"loop_1iter": dict(two=False, task={}), "loop_1iter_two": dict(two=True, task={}),
                 "loop_1iter_time": dict(two=False, task=dict(max_episode_length=0.4)),
                 # task.num_disc_obs_steps = 2: two-deep history ring, 76-wide discriminator input
                 "loop_1iter_s2": dict(two=False, task=dict(num_disc_obs_steps=2)),
                 "loop_1iter_s4": dict(two=False, task=dict(num_disc_obs_steps=4))}


@pytest.mark.parametrize("name,precision", [("loop_1iter", p) for p in PRECISIONS] + [("loop_1iter_two", "fp32"), ("loop_1iter_time", "fp32"), ("loop_1iter_s2", "fp32"), ("loop_1iter_s4", "fp32")])
def test_one_full_iteration_matches_reference_and_oracle(name, precision):
    """BASELINE config 1 stand-in: the reference's own iteration (fake kinematic engine, recorded draws) replayed
    through the HIP engine.  Variants: the two-clip library (clip draws, raw-frame table offsets, [2,20] sampler table) and a
    0.4 s episode limit with pre-aged episode clocks, whose DONE_TIME samples must bootstrap from the PRE-reset observation
    (ppo_agent.py:117-133; here: the obs_timeout rows, since no next_obs buffer is kept)."""
    import torch
    from oracle import loop as LP
    from oracle import task as OT
    from tests.util import oracle_lib

    g = gload(name)
    var = LOOP_VARIANTS[name]
    Tn, n = g["noise"].shape[:2]
    cfg = make_cfg(n, steps_per_iter=Tn, matmul_precision=precision)
    cfg["task"].update(var["task"])
    ms = gload("motion_small")
    frames, weights = ([ms["two_frames0"], ms["two_frames1"]], [1.0, 3.0]) if var["two"] else ([ms["frames"]], [1.0])
    ag = make_agent(cfg, frames, weights)
    dd = 38 * var["task"].get("num_disc_obs_steps", 3)
    ag._model.load({k: torch.tensor(v) for k, v in OL.synth_params(int(g["seed"]), disc_dim=dd).items()})

    # ---- oracle run alongside: supplies the per-step sampler probabilities needed to turn the reference's
    # multinomial draws into the uniforms of the device sampler, and a second opinion on every output
    lib = oracle_lib(two=var["two"], golden_tables=True)
    orc = LP.Agent(LP.AgentCfg(), OT.TaskCfg(**var["task"]), lib, n, OL.synth_params(int(g["seed"]), disc_dim=dd))
    clip_cdf = np.cumsum(lib.weights, dtype=F)
    init = dict(ids=g["init_ids"], segments=g["init_segments"], jitter=g["init_jitter"])
    resets = []
    for t in range(Tn):
        k = int(g["reset_count"][t])
        resets.append(dict(ids=g["reset_ids"][t, :k], segments=g["reset_segments"][t, :k], jitter=g["reset_jitter"][t, :k]))
    plan = g["contact_plan"]

    def uniforms(env_ids, draw):
        u = np.zeros((3, n), F)
        probs = orc.task.sampler.probs(draw["ids"]) if len(env_ids) else None
        for j, e in enumerate(env_ids):
            u[0, e] = _mid(clip_cdf, int(draw["ids"][j]))  # the clip the reference's multinomial drew
            u[1, e] = _mid(np.cumsum(probs[j]), int(draw["segments"][j]))
            u[2, e] = draw["jitter"][j]
        return T(u)

    inj_u = {ag.stream_reset_all(0): uniforms(np.arange(n), init)}
    orc.init(init)
    preset = g["time_preset"].astype(F)  # zeros except in the time variant: episode clocks aged, motion clocks untouched
    orc.task.time = (orc.task.time + preset).astype(F)
    orc.task.time_off = (orc.task.time_off - preset).astype(F)
    errors0 = orc.task.sampler.errors.copy()  # the table in effect during the whole rollout (it is updated in build-train-data)
    orc_info = orc.train_iter(LP.Draws(g["noise"], resets, g["perms"]), [(plan[t] >= 0) for t in range(Tn)])
    errors1, orc.task.sampler.errors = orc.task.sampler.errors, errors0
    for t in range(Tn):
        ids = np.nonzero(orc.buf["done"][t] != 0)[0]
        inj_u[ag.stream_train_reset(0 * Tn + t)] = uniforms(ids, resets[t])
    orc.task.sampler.errors = errors1
    assert np.all(g["reset_count"] == [(orc.buf["done"][t] != 0).sum() for t in range(Tn)])

    ent = ag._env.robot.entity
    contact_flags = T((plan >= 0).astype(np.uint8))

    def pre_step(t):
        ent.forced_contact.copy_(contact_flags[t])

    perms = iter([torch.tensor(p) for p in g["perms"]])
    ag.inject = dict(noise=T(g["noise"]), uniforms=inj_u, perms=perms, pre_step=pre_step)
    ag.reset_all_envs()
    ag._init_train()
    ag._S["time"].add_(T(preset))
    ag._S["time_off"].sub_(T(preset))
    info = ag._train_iter()
    torch.cuda.synchronize()
    B = ag._B
    done = B["done"].cpu().numpy()
    assert np.array_equal(done, g["buf.done"])                                   # bit-exact flags, whole rollout
    if name == "loop_1iter_time":
        assert (done == 3).sum() >= 10
    if name == "loop_1iter_two":
        assert set(np.unique(B["motion_id"].cpu().numpy())) == {0, 1}
    assert np.array_equal(B["motion_time"].cpu().numpy(), g["buf.motion_times"])  # bit-exact clocks and reset times
    np.testing.assert_allclose(B["obs"][Tn - 1].cpu().numpy()[:, :264], g["buf.obs_last"], rtol=0, atol=5e-5)
    assert np.all(B["obs"].cpu().numpy()[..., 264:] == 0)
    np.testing.assert_allclose(B["reward"].cpu().numpy(), g["buf.reward"], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(B["adv"].cpu().numpy(), g["buf.adv"], rtol=5e-3, atol=5e-3)
    np.testing.assert_allclose(B["adv"].cpu().numpy(), orc.buf["adv"], rtol=5e-3, atol=5e-3)
    # (no next_obs buffer here: V(next_obs[t]) is V(obs[t+1]) except for reset envs -- covered through tar_val / adv)
    for k, key in (("obs", "obs"), ("a_logp", "a_logp"), ("tar_val", "tar_val")):
        ref = float(g[f"buf.{k}.abs"])
        got = B[key][:Tn].double().abs().sum().item()
        assert abs(got - ref) <= 5e-5 * ref, (k, got, ref)
    for k in ("adv_mean", "adv_std", "disc_reward_mean", "disc_reward_std", "loss", "actor_loss", "critic_loss", "disc_loss", "clip_frac", "imp_ratio",
              "disc_grad_penalty", "disc_logit_loss", "disc_pos_acc", "disc_neg_acc", "disc_pos_logit", "disc_neg_logit", "mean_return", "mean_ep_len",
              "num_eps"):
        np.testing.assert_allclose(info[k], float(g["info." + k]), rtol=1e-2, atol=5e-4, err_msg=k)
        np.testing.assert_allclose(info[k], orc_info[k], rtol=1e-2, atol=5e-4, err_msg="oracle:" + k)
    Nm = ag._Nrm
    np.testing.assert_allclose(Nm["obs_mean"][:264].cpu().numpy(), g["obs_mean"], rtol=1e-4, atol=1e-5)
    scale = g["obs_mean"] ** 2 + g["obs_std"] ** 2
    assert np.all(np.abs(Nm["obs_std"][:264].cpu().numpy() ** 2 - g["obs_std"] ** 2) <= 5e-6 * scale + 1e-9)
    np.testing.assert_allclose(Nm["d_abs"][:dd].cpu().numpy(), g["disc_mean_abs"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ag._smp["errors"].cpu().numpy(), g["sampler_errors"], rtol=1e-4)
    ph = {k: v.numpy() for k, v in ag._model.export().items() if k != "_model._action_dist._logstd_net"}
    # (the actor's first layer is where 40 Adam steps amplify rounding most -- mean |difference| of the sampled weights over the fixtures, fp32:
    # 1.2e-5, 2.8e-5, 3.0e-5, two-step fixture 5.9e-5, of 4e-3 travelled -- while critic and discriminator stay within 1e-9 .. 3e-7 in all four,
    # the two-step discriminator included; the two-step fixture gets twice the mean bound)
    _check_param_summary(g, "param", ph, 40, long_run=True, mean_frac=0.02 if name in ("loop_1iter_s2", "loop_1iter_s4") else 0.01)


def test_checkpoint_roundtrip_uses_reference_keys(tmp_path):
    import torch

    cfg = make_cfg(16, steps_per_iter=4)
    ag = make_agent(cfg, [gload("motion_small")["frames"]], [1.0])
    ag._model.opt_step = 3
    ag._model.exp_avg.normal_()
    path = str(tmp_path / "model.pt")
    ag.save(path)
    ck = torch.load(path, weights_only=True)
    assert set(ck) == {"model", "optimizer", "iter", "sample_count"}
    names = {n for n, _ in OL.PARAM_SHAPES}
    assert names <= set(ck["model"])
    for n, shape in OL.PARAM_SHAPES:
        assert tuple(ck["model"][n].shape) == shape, n
    assert {"_obs_norm._count", "_obs_norm._mean", "_obs_norm._std", "_a_norm._mean", "_disc_obs_norm._mean_abs", "_model._action_dist._logstd_net"} <= set(ck["model"])
    assert len(ck["optimizer"]["state"]) == 22
    # ... and against the reference agent's own checkpoint (fixture generated from the imported reference): every tensor name,
    # shape and dtype of state_dict(), the optimizer state layout and its param-group keys
    import json

    meta = json.loads(bytes(gload("state_dict")["meta"]).decode())
    assert [k for k, _, _ in meta["model"]] == list(ck["model"].keys())  # same names in the same order
    for k, shape, dt in meta["model"]:
        assert list(ck["model"][k].shape) == shape and str(ck["model"][k].dtype) == dt, k
    for i, st in meta["opt_state"]:
        got = ck["optimizer"]["state"][i]
        assert set(got) == set(st) and list(got["exp_avg"].shape) == st["exp_avg"] and list(got["exp_avg_sq"].shape) == st["exp_avg_sq"], i
        assert got["step"].shape == torch.Size([])
    assert set(ck["optimizer"]["param_groups"][0]) == set(meta["opt_group"])
    assert sorted(ck) == meta["top"] and type(ck["iter"]).__name__ == meta["iter_type"] and type(ck["sample_count"]).__name__ == meta["sample_count_type"]
    before = ag._model.params.clone()
    ag._model.params.zero_()
    ag.load(path)
    assert torch.equal(ag._model.params, before) and ag._model.opt_step == 3
    # DDP-prefixed checkpoints load too (base_agent.py:190-203)
    ck["model"] = {k.replace("_model.", "_model.module."): v for k, v in ck["model"].items()}
    torch.save(ck, path)
    ag._model.params.zero_()
    ag.load(path)
    assert torch.equal(ag._model.params, before)


@pytest.mark.parametrize("std_type,precision", [("CONSTANT", "fp32"), ("VARIABLE", "fp32"), ("VARIABLE", "bf16"), ("CONSTANT", "bf16")])
def test_trainable_std_types_train_and_checkpoint(tmp_path, std_type, precision):
    """actor_std_type CONSTANT / VARIABLE through the whole loop: two iterations (graph rollout on) stay finite and move the log-std
    parameters; the checkpoint carries the reference's keys for that type in the reference's registration order (a Parameter of the
    distribution module before its sub-modules: distribution_gaussian_diag.py:19-43), the optimiser state one entry more / two more than
    FIXED, and loads back bit for bit."""
    import torch
    import add_gym_amd.learning.add_agent as A

    cfg = make_cfg(256, steps_per_iter=8, matmul_precision=precision, rollout_graph=True)
    cfg["agent"]["model"]["actor_std_type"] = std_type
    cfg["task"]["motion_file"] = "synthetic:2x300"
    ag = A.ADDAgent(cfg)
    ag.reset_all_envs()
    ag._init_train()
    key = "_model._action_dist._logstd_net" + (".bias" if std_type == "VARIABLE" else "")
    before = ag._model.export()[key].clone()
    assert torch.allclose(before, torch.full_like(before, float(np.log(0.05))))
    for _ in range(2):
        info = ag._train_iter()
        ag._iter += 1
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in info.values())
    after = ag._model.export()
    assert float((after[key] - before).abs().max()) > 1e-4  # 80 optimiser steps moved it
    path = str(tmp_path / "model.pt")
    ag.save(path)
    ck = torch.load(path, weights_only=True)
    names = list(ck["model"])
    i_mean = names.index("_model._action_dist._mean_net.weight")
    if std_type == "CONSTANT":
        assert names[i_mean - 1] == "_model._action_dist._logstd_net" and tuple(ck["model"][names[i_mean - 1]].shape) == (29,)
        assert len(ck["optimizer"]["state"]) == 23
    else:
        assert names[i_mean + 2:i_mean + 4] == ["_model._action_dist._logstd_net.weight", "_model._action_dist._logstd_net.bias"]
        assert tuple(ck["model"][names[i_mean + 2]].shape) == (29, 512) and tuple(ck["model"][names[i_mean]].shape) == (29, 512)
        assert len(ck["optimizer"]["state"]) == 24
    params = ag._model.params.clone()
    ag._model.params.zero_()
    ag.load(path)
    assert torch.equal(ag._model.params, params)


def test_training_runs_and_learns_signal(tmp_path):
    """Two iterations of train_model at a small size through the public surface (log file keys, checkpoint written)."""
    cfg = make_cfg(256, steps_per_iter=8, iters_per_output=1, test_episodes=0, max_samples=2 * 8 * 256)
    cfg["task"]["motion_file"] = "synthetic:2x300"
    import add_gym_amd.learning.add_agent as A

    ag = A.ADDAgent(cfg)
    out = tmp_path / "model.pt"
    ag.train_model(str(out), str(tmp_path), str(tmp_path / "log.txt"))
    assert out.exists()
    header = (tmp_path / "log.txt").read_text().splitlines()[0].split()
    for k in ("Iteration", "Wall_Time", "Samples", "Test_Return", "Train_Return", "Loss", "Critic_Loss", "Actor_Loss", "Clip_Frac", "Imp_Ratio",
              "Disc_Loss", "Disc_Grad_Penalty", "Disc_Logit_Loss", "Disc_Pos_Acc", "Disc_Neg_Acc", "Adv_Mean", "Adv_Std", "Disc_Reward_Mean", "Exp_Prob"):
        assert k in header, k
    assert ag._sample_count == 2 * 8 * 256


def test_bf16_storage_rollout_trains_like_the_default_rollout():
    """agent.matmul_precision=bf16 with agent.rollout_precision=bf16_storage (rollout / value / discriminator-reward passes on bf16 storage,
    inside the hipGraph rollout too): two iterations from the same seed stay close to the default bf16x2 rollout in every logged scalar
    that the 8-bit operands can move only a little, and the first iteration's flags / episode counts are identical."""
    import torch
    import add_gym_amd.learning.add_agent as A

    infos = {}
    for roll in (None, "bf16_storage"):
        cfg = make_cfg(512, steps_per_iter=16, matmul_precision="bf16", **({"rollout_precision": roll} if roll else {}))
        cfg["task"]["motion_file"] = "synthetic:2x300"
        ag = A.ADDAgent(cfg)
        assert ag._roll_storage == (roll is not None)
        ag.reset_all_envs()
        ag._init_train()
        infos[roll] = [ag._train_iter() for _ in range(2)]
        torch.cuda.synchronize()
        assert all(np.isfinite(v) for it in infos[roll] for v in it.values())
    a, b = infos[None][0], infos["bf16_storage"][0]
    assert a["num_eps"] == b["num_eps"] and a["mean_ep_len"] == b["mean_ep_len"]
    for k in ("disc_reward_mean", "critic_loss", "disc_loss", "adv_std"):
        np.testing.assert_allclose(b[k], a[k], rtol=5e-2, atol=5e-3, err_msg=k)


def test_foreign_engine_slow_path_matches_in_place_path():
    """An engine that only offers the reference's BaseEntity getters/setters gives the same rollout as the in-place one."""
    import torch

    outs = []
    for target in ("add_gym_amd.engine.kinematic_engine.KinematicEngine", "tests.foreign_engine.ForeignEngine"):
        cfg = make_cfg(128, steps_per_iter=8)
        cfg["engine"]["_target_"] = target
        cfg["task"]["motion_file"] = "synthetic:2x45"  # 1.5 s clips: several of the 128 envs run past the clip end within 8 steps
        torch.manual_seed(0)
        import add_gym_amd.learning.add_agent as A

        ag = A.ADDAgent(cfg)
        assert ag._fast_engine == (target.endswith("KinematicEngine"))
        ag.reset_all_envs()
        ag._init_train()
        info = ag._train_iter()
        torch.cuda.synchronize()
        outs.append((ag._B["done"].clone(), ag._B["reward"].clone(), ag._B["obs"][:8].clone(), ag._B["motion_time"].clone(), info))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
    assert torch.equal(a[2], b[2])
    torch.testing.assert_close(a[1], b[1], rtol=0, atol=0)
    assert int((a[0] != 0).sum()) > 0  # resets happened, so set_qpos / set_dofs_velocity were exercised


def test_test_model_runs_episode_quota_with_deterministic_policy():
    """Corrected mode (task.reference_compat=false): fresh reset of all envs, ceil(n / num_envs) episodes per env."""
    cfg = make_cfg(64, steps_per_iter=8)
    cfg["task"].update(motion_file="synthetic:1x90", reference_compat=False)  # 3 s clip -> episodes end (SUCC) within 300 steps
    import add_gym_amd.learning.add_agent as A

    ag = A.ADDAgent(cfg)
    info = ag.test_model(128)  # ceil(128/64) = 2 episodes per env
    assert info["num_eps"] >= 128
    assert 0 < info["mean_ep_len"] <= 300 and np.isfinite(info["mean_return"])
    assert ag._mode == A.AgentMode.TRAIN


def test_test_model_matches_the_reference_rollout():
    """BaseAgent.test_model / _rollout_test (base_agent.py:116-126, 393-425) replayed with the reference's recorded reset draws
    (fixture test_rollout.npz, generated from the imported reference on its fake engine): the same number of steps, bit-identical
    done flags for every env and step, per-step rewards, mean return / episode length / episode count.  Includes the
    reference's TEST-mode behaviour: only env 0 is reset at the start, every env runs `num_episodes` episodes."""
    import torch

    g = gload("test_rollout")
    n, episodes, steps = int(g["num_envs"]), int(g["episodes"]), int(g["steps"])
    cfg = make_cfg(n, steps_per_iter=8)
    ag = make_agent(cfg, [gload("motion_small")["frames"][:int(g["clip_frames"])]], [1.0])
    ag._model.load({k: torch.tensor(v) for k, v in OL.synth_params(int(g["seed"])).items()})
    probs = np.full(20, 1.0 / 20, F)  # sampler errors are all ones: uniform segment probabilities (sampler.py:57-73)
    cdf = np.cumsum(probs)

    def uniforms(env_ids, segs, jit):
        u = np.zeros((3, n), F)
        for j, e in enumerate(env_ids):
            u[0, e], u[1, e], u[2, e] = 0.5, _mid(cdf, int(segs[j])), jit[j]
        return T(u)

    done_ref = g["done"]
    inj = {ag.stream_reset_all(0): uniforms(np.arange(n), g["init_segments"], g["init_jitter"]),
           ag.stream_test_reset_all(0): uniforms([0], g["first_segments"], g["first_jitter"])}
    for t in range(steps):
        k = int(g["reset_count"][t])
        inj[ag.stream_test_reset(0, t)] = uniforms(np.nonzero(done_ref[t] != 0)[0], g["reset_segments"][t, :k], g["reset_jitter"][t, :k])
    for t in range(steps, steps + 64):  # (only reached if the rollout ran longer than the reference's: fails below)
        inj[ag.stream_test_reset(0, t)] = uniforms([], [], [])
    dones, ep = [], {}
    # record done flags per step: the reset kernel clears them, so snapshot between the step and the reset
    orig_reset = ag._reset_envs

    def reset_spy(reset_all, *a):
        if not reset_all and len(dones) < steps + 64 and ep.get("live"):
            dones.append(ag._S["done"].clone())
        return orig_reset(reset_all, *a)

    ag._reset_envs = reset_spy
    ag.inject = dict(uniforms=inj)
    ag.reset_all_envs()      # train_model's reset before the loop (TRAIN mode: all envs)
    ag._init_train()
    ep["live"] = False
    # the first masked reset inside test_model is the "reset env 0" one: start recording after it
    orig_decide = ag._decide_action

    def decide_spy(*a):
        ep["live"] = True
        return orig_decide(*a)

    ag._decide_action = decide_spy
    info = ag.test_model(episodes)
    torch.cuda.synchronize()
    got = torch.stack(dones).cpu().numpy()
    assert ag._test_steps == steps
    assert np.array_equal(got, done_ref)
    assert info["num_eps"] == int(g["num_eps"])
    np.testing.assert_allclose(info["mean_ep_len"], float(g["mean_ep_len"]), rtol=1e-6)
    np.testing.assert_allclose(info["mean_return"], float(g["mean_return"]), rtol=2e-5)


def test_multi_clip_library_and_corrected_offsets():
    import torch
    import add_gym_amd.learning.add_agent as A

    for compat in (True, False):
        cfg = make_cfg(256, steps_per_iter=8)
        cfg["task"].update(motion_file="synthetic:3x150", reference_compat=compat)
        ag = A.ADDAgent(cfg)
        ag.reset_all_envs()
        ag._init_train()
        info = ag._train_iter()
        ids = ag._S["motion_id"].cpu().numpy()
        assert set(np.unique(ids)) == {0, 1, 2}
        assert all(np.isfinite(v) for v in info.values())
        lib = ag._motion_lib
        assert torch.isfinite(ag._B["obs"]).all() and torch.isfinite(ag._B["disc_demo"]).all()
        if not compat:  # corrected mode: each env's reference row belongs to its own clip
            t = ag._S["time"] + ag._S["time_off"]
            idx = lib.step_index(ag._S["motion_id"], t).cpu().numpy()
            start, cnt = lib._step_start.numpy(), lib._step_counts.numpy()
            assert np.all((idx >= start[ids]) & (idx < start[ids] + cnt[ids]))


def test_velocity_and_phase_observations_train_end_to_end():
    """task.enable_vel_obs / enable_phase_obs widen obs to 308 and disc obs to 219 columns; the whole iteration
    (rollout, velocity history ring, discriminator, update) runs on those widths.  Per-element parity of the wider
    rows is pinned in test_hip_env (golden variants vel_phase / local_vel)."""
    import torch
    import add_gym_amd.learning.add_agent as A

    cfg = make_cfg(128, steps_per_iter=8)
    cfg["task"].update(motion_file="synthetic:2x240", enable_vel_obs=True, enable_phase_obs=True, num_phase_encoding=4)
    ag = A.ADDAgent(cfg)
    assert (ag._task.obs_dim, ag._task.disc_dim) == (264 + 35 + 9, 3 * (38 + 35))
    ag.reset_all_envs()
    ag._init_train()
    info = ag._train_iter()
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in info.values())
    # the velocity ring's newest slot holds the simulator velocities of the last step
    # (for envs reset at the end of the step: the clip velocities the reset wrote to both places)
    hv = ag._S["hist_vel"]
    assert torch.isfinite(hv).all() and float(hv.abs().sum()) > 0
    assert torch.equal(hv[:, (ag._head - 1) % 3], ag._S["sim_vel"])


def test_exploration_probability_annealing_masks_samples():
    """agent.exp_prob_beg / exp_prob_end / exp_anneal_samples (ppo_agent.py:32-34, 80-88, 161-168): a share of the envs takes the
    mode each step, those samples carry rand_action_mask 0 and stay out of the actor loss and the advantage statistics."""
    import torch
    import add_gym_amd.learning.add_agent as A

    cfg = make_cfg(256, steps_per_iter=8, exp_prob_beg=0.5, exp_prob_end=0.25, exp_anneal_samples=4 * 8 * 256)
    cfg["task"]["motion_file"] = "synthetic:2x240"
    ag = A.ADDAgent(cfg)
    ag.reset_all_envs()
    ag._init_train()
    assert ag._get_exp_prob() == 0.5
    info = ag._train_iter()
    ag._sample_count = ag._total_samples
    torch.cuda.synchronize()
    frac = float(ag._B["rand_mask"][:8].mean())
    assert 0.4 < frac < 0.6 and all(np.isfinite(v) for v in info.values())
    assert abs(ag._get_exp_prob() - 0.4375) < 1e-9  # a quarter of the way from 0.5 to 0.25
    # mode actions: a_logp equals the density at the mode wherever the mask is 0
    m0 = ag._B["rand_mask"][:8] == 0
    assert torch.allclose(ag._B["a_logp"][:8][m0], torch.tensor(ag._model.logp_const, device="cuda"))


def test_kinematic_playback_export(tmp_path):
    """add_gym_amd.view: the GUI-less playback export; with view.source=reference the character is placed on the clip every
    step, so simulator and reference poses coincide and the imitation reward is at its maximum."""
    from add_gym_amd.view import export_playback

    cfg = make_cfg(8, steps_per_iter=8)
    cfg["task"]["motion_file"] = "synthetic:1x300"
    out = tmp_path / "playback.npz"
    arrays = export_playback(cfg, str(out), 20, source="reference")
    z = np.load(out, allow_pickle=False)
    assert z["sim_pose"].shape == (20, 8, 36) and z["ref_pose"].shape == (20, 8, 36) and len(z["body_names"]) == 30
    live = arrays["done"] == 0  # (envs that ran past the clip end are reset after the step)
    assert live.mean() > 0.5 and np.abs(arrays["sim_pose"] - arrays["ref_pose"])[live].max() < 1e-3  # fp32 clock vs recomputed time
    assert arrays["reward"][live].min() > 0.84  # all four reward terms at their maximum (0.5 + 0.1 + 0.15 + 0.1)
    # policy playback: every exported reward and done flag recomputed by the CPU oracle from the exported simulator / reference
    # states (add_reward.py:103-177, add_done.py:96-147)
    from oracle import task as OT

    cfg2 = make_cfg(16, steps_per_iter=8)
    cfg2["task"]["motion_file"] = "synthetic:1x60"  # 2 s clip: some envs run past its end (SUCC), the random policy fails poses (FAIL)
    arr = export_playback(cfg2, str(out), 60, source="policy")
    tcfg = OT.TaskCfg()
    unpack = lambda p, v: (p[:, 0:3], p[:, 3:7], v[:, 0:3], v[:, 3:6], p[:, 7:36], v[:, 6:35])
    clip_len = np.float32((60 - 1) / 30.0)
    n_done = 0
    for k in range(60):
        sim, ref = unpack(arr["sim_pose"][k], arr["sim_vel"][k]), unpack(arr["ref_pose"][k], arr["ref_vel"][k])
        np.testing.assert_allclose(arr["reward"][k], OT.reward(tcfg, sim, ref), rtol=0, atol=5e-6, err_msg=f"step {k}")
        want = OT.done_flags(tcfg, arr["time"][k], arr["motion_time"][k], np.full(16, clip_len, np.float32), np.zeros(16, np.int32), sim[0], sim[4], ref[0], ref[4], None)
        assert np.array_equal(arr["done"][k], want), k
        n_done += int((want != 0).sum())
    assert n_done > 0


def test_episode_time_limit_rows_feed_the_critic():
    """DONE_TIME samples: the step kernel parks the pre-reset obs row in obs_timeout, the critic evaluates those rows, and
    TD(lambda) bootstraps from that value (ppo_agent.py:117-133) -- not from V(obs[t+1]), which is the reset obs."""
    import torch
    import add_gym_amd.learning.add_agent as A

    cfg = make_cfg(256, steps_per_iter=8)
    cfg["task"].update(motion_file="synthetic:1x600", enable_early_termination=False, max_episode_length=0.2)  # 20 steps, clip 20 s
    ag = A.ADDAgent(cfg)
    ag.reset_all_envs()
    ag._init_train()
    hits = 0
    for it in range(3):
        ag._B["obs"][0].copy_(ag._B["obs"][ag.T]) if it else None
        ag._rollout_train()
        ag._build_train_data()
        ag._iter += 1
        torch.cuda.synchronize()
        B = ag._B
        done, r, tar = B["done"][-1], B["reward"][-1], B["tar_val"][-1]   # last step of the rollout: ret = r + gamma * next value
        t_env = (done == 3).nonzero().flatten()
        hits += int((B["done"] == 3).sum())
        if len(t_env):
            want = r[t_env] + ag._discount * B["timeout_vals"][t_env]
            torch.testing.assert_close(tar[t_env], want, rtol=1e-6, atol=1e-6)
            other = r[t_env] + ag._discount * B["vals"][ag.T][t_env]   # what bootstrapping from the reset obs would give
            assert (want - other).abs().max() > 1e-4
            # the parked row is the observation of that step before the reset overwrote slot T
            assert not torch.equal(B["obs_timeout"][t_env], B["obs"][ag.T][t_env])
        assert torch.isfinite(B["tar_val"]).all() and torch.isfinite(B["adv"]).all() and torch.isfinite(B["timeout_vals"]).all()
    assert hits >= 200  # nearly every env ran into the 0.2 s limit in 24 steps (a few reach the clip end first: DONE_SUCC)


def test_rollout_graph_replays_the_same_rollout():
    """agent.rollout_graph: the T rollout steps of an iteration captured into one hipGraph per ring phase (BASELINE configs[4]:
    "hipGraph-captured rollout step").  Replays must draw the same Philox numbers as the call-by-call path and produce
    bit-identical buffers, iteration after iteration (eager warm-up, capture, then pure replays, all three ring phases)."""
    import torch
    import add_gym_amd.learning.add_agent as A

    bufs = []
    for graph in (False, True):
        cfg = make_cfg(512, steps_per_iter=8, rollout_graph=graph)
        cfg["task"]["motion_file"] = "synthetic:2x60"
        cfg["seed"] = 4
        ag = A.ADDAgent(cfg)
        ag.reset_all_envs()
        ag._init_train()
        out = []
        for it in range(7):
            if it:
                ag._B["obs"][0].copy_(ag._B["obs"][ag.T])
            ag._rollout_train()
            ag._iter += 1
            torch.cuda.synchronize()
            out.append({k: ag._B[k].clone() for k in ("obs", "action", "a_logp", "done", "reward", "motion_time", "disc_obs", "ep_stats")})
        bufs.append(out)
        if graph:
            assert len(ag._graphs) == 3 and ag._total_samples == 7 * 8 * 512  # T = 8: the ring phase advances by 2 per iteration
    for it, (a, b) in enumerate(zip(*bufs)):
        for k in a:
            if k == "ep_stats":  # sums of finished returns / lengths accumulated by float atomics: order varies run to run
                torch.testing.assert_close(a[k], b[k], rtol=1e-5, atol=1e-5)
            else:
                assert torch.equal(a[k], b[k]), (it, k)
    assert int((bufs[0][-1]["done"] != 0).sum()) > 0

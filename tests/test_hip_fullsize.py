"""Full-size agent-level parity on the GPU (BASELINE configs[1] and configs[2] shapes):

* the recorded update plan of a 4096-env agent (16 384-row minibatch: split-K slabs, ReLU sign bits, the three-stream
  schedule, the Mb+1 row-chunk split) run once, all 22 parameter gradients against the CPU oracle's autograd
  (oracle.learn.compute_loss; restates ppo_agent.py:171-275 and add_agent.py:141-202), under every matmul precision;
* one rollout + build-train-data of a 16 384-env agent on a 5-clip library under bf16x3, a 256-env subset compared with
  oracle/loop.py run on exactly those envs with the same draws (obs / reward / done / clocks / TD(lambda) targets).
"""
import numpy as np
import pytest

from oracle import learn as OL
from tests.test_hip_agent import T, make_cfg, sync_minibatch16

pytestmark = pytest.mark.gpu
F = np.float32

# Gradient tolerances.  The yardstick is the SAME minibatch evaluated in float64 (oracle.learn.float64_mode): at 16 384 rows the
# gradients are sums with heavy cancellation, and torch-CPU's own fp32 autograd is 1e-5 .. 2e-3 away from it (relative L2 per tensor,
# profiles/r02_grad_error_table.log).  fp32 MFMA and the exact bf16x3 split must be as close to float64 as torch's fp32 is, give or
# take summation order (factor 8, floor 2e-4); bf16x2 keeps 16 significant bits per operand (2^-15 |a||b| per product).
# f16x2 keeps 22 bits + sign per operand on per-tensor scales (ADDHIP_PREC_F16X2): the same rule with twice the floor -- its worst tensor
# (a hidden-layer weight gradient of the actor, heavy cancellation) sits at 2.5e-4 where the fp32 MFMA sits at 1.2e-4
# (profiles/r04_grad_error_table.log).
L2_FACTOR, L2_FLOOR = {"fp32": 8.0, "bf16x3": 8.0, "bf16x3_planes": 8.0, "f16x2": 8.0}, {"f16x2": 4e-4}
L2_ABS = {"bf16x2": 1.5e-2, "bf16": 1.2e-1}   # bf16 = bf16 STORAGE of activations / gradients / weight shadow (8 significant bits, nearest even)
MAX_ABS = {"fp32": 4e-2, "bf16x3": 4e-2, "bf16x3_planes": 4e-2, "f16x2": 4e-2, "bf16x2": 1e-1, "bf16": 4e-1}   # largest element error / largest gradient element, per tensor (torch-CPU fp32 itself: up to 3e-2)


def _fill_minibatch(ag, model, seed):
    """A plausible minibatch written straight into the agent's gathered-minibatch buffers; returns the oracle's view."""
    rng = np.random.RandomState(seed)
    M = ag.Mb
    obs = rng.standard_normal((M, 264)).astype(F)
    import torch

    with torch.no_grad():
        mean = model.actor_mean(OL.t32(obs))
        act = (mean + 0.05 * torch.tensor(rng.standard_normal((M, 29)).astype(F))).numpy()
        logp = model.log_prob(mean, OL.t32(act)).numpy()
    # old log-probabilities: offsets from the current ones that keep every importance ratio well away from the PPO clip
    # boundaries 0.8 / 1.2 (the clipped objective's gradient is discontinuous there: a sample within rounding of a boundary
    # would make ANY two implementations disagree by that sample's whole contribution)
    off = rng.choice(np.array([-0.4, -0.1, 0.05, 0.1, 0.3], F), M)
    mb = dict(norm_obs=obs, norm_action=act, a_logp=(logp + off).astype(F),
              adv=np.clip(rng.standard_normal(M), -4, 4).astype(F), tar_val=rng.standard_normal(M).astype(F),
              rand_action_mask=(rng.rand(M) < 0.95).astype(F), norm_diff=(0.5 * rng.standard_normal((M, 114))).astype(F))
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
    return mb


_ORACLE_GRADS = {}  # CPU autograd gradients of the minibatch, shared by the precision parametrisation


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16x3_planes", "f16x2", "bf16x2", "bf16"])
def test_update_plan_at_4096_envs_all_gradients_match_oracle(precision):
    import torch
    import add_gym_amd.learning.add_agent as A

    cfg = make_cfg(4096, steps_per_iter=32, matmul_precision=precision)
    cfg["task"]["motion_file"] = "synthetic:1x300"
    ag = A.ADDAgent(cfg)
    assert ag.Mb == 16384
    seed = 11
    params = OL.synth_params(seed)
    ag._model.load({k: torch.tensor(v) for k, v in params.items()})
    if hasattr(ag._model, "refresh_shadow"):
        ag._model.refresh_shadow()
    torch.set_num_threads(16)
    model = OL.Model(params)
    mb = _fill_minibatch(ag, model, 5)
    ag._W["stats"].zero_()
    ag._run_update_sections()  # zero_grad + the actor / critic / discriminator sections on their three streams
    torch.cuda.synchronize()
    m = ag._model
    grads_hip = {k: v.numpy() for k, v in m.export(m.grads).items() if k != "_model._action_dist._logstd_net"}
    key = ("grads", seed)
    if key not in _ORACLE_GRADS:
        names = model.names()
        loss, info = OL.compute_loss(model, OL.LossCfg(), mb)
        g32 = {n: g.numpy().astype(np.float64) for n, g in zip(names, torch.autograd.grad(loss, [model.p[n] for n in names]))}
        with OL.float64_mode():
            m64 = OL.Model(params)
            loss64, _ = OL.compute_loss(m64, OL.LossCfg(), mb)
            g64 = {n: g.numpy() for n, g in zip(names, torch.autograd.grad(loss64, [m64.p[n] for n in names]))}
        _ORACLE_GRADS[key] = (g32, g64, info)
    g32, g64, info = _ORACLE_GRADS[key]
    assert set(g64) == set(grads_hip) and len(g64) == 22
    for k, ref in g64.items():
        gh = grads_hip[k].astype(np.float64)
        l2 = np.linalg.norm(ref) + 1e-300
        err = np.linalg.norm(gh - ref) / l2
        err32 = np.linalg.norm(g32[k] - ref) / l2
        bound = L2_ABS[precision] if precision in L2_ABS else max(L2_FACTOR[precision] * err32, L2_FLOOR.get(precision, 2e-4))
        assert err <= bound, (precision, k, float(err), float(err32))
        emax = np.abs(gh - ref).max() / (np.abs(ref).max() + 1e-300)
        assert emax <= MAX_ABS[precision], (precision, k, float(emax))
    # logged scalars of the step
    ag._total_samples = ag.T * ag.N
    stats = ag._collect_info(1)
    tol = 2e-2 if precision == "bf16" else 1e-3
    for k in ("critic_loss", "actor_loss", "disc_loss", "disc_grad_penalty", "disc_logit_loss", "disc_neg_logit", "clip_frac", "imp_ratio"):
        np.testing.assert_allclose(stats[k], info[k], rtol=tol, atol=tol, err_msg=k)
    # padded rows / columns of the device layout never receive gradient
    assert float(m.view("actor", "Wh", m.grads)[29:].abs().max()) == 0
    assert float(m.view("disc", "W0", m.grads)[:, 114:].abs().max()) == 0
    assert float(m.view("actor", "W0", m.grads)[:, 264:].abs().max()) == 0


# BASELINE configs[2] tolerances per matmul mode (abs unless noted), against the fp32 CPU oracle on the same draws.  bf16x3 forms
# fp32-exact products.  With agent.matmul_precision=bf16 the rollout / evaluation passes run bf16x2 products on fp32 operands (16
# significant bits per operand, 2^-15 |a||b| per product; bf16 STORAGE applies to the update step only), so actions carry ~1e-4 of
# product error at most, the kinematic stand-in feeds it back into the next observation, and the value / logit heads (sums over 512
# units of O(1) activations) carry up to ~1e-3; the integer outcomes (done flags, clocks, clip ids, reset start times) must not move at all.
CFG2_TOL = {"bf16x3": dict(obs=5e-5, action=2e-5, logp_r=1e-4, logp_a=1e-3, reward_r=2e-3, reward_a=2e-4, tar_r=2e-3, tar_a=2e-3),
            "bf16": dict(obs=5e-5, action=5e-5, logp_r=1e-4, logp_a=1e-3, reward_r=5e-3, reward_a=5e-4, tar_r=5e-3, tar_a=3e-3),
            # agent.rollout_precision = bf16_storage: the rollout / evaluation passes on bf16 storage too (operands rounded to 8 bits)
            # (every GEMM operand of the actor / critic / discriminator passes rounded to 8 significant bits: the action means move by a few
            # 1e-4, the kinematic stand-in carries that into the observations, the 512-unit value / logit sums by ~1e-2; flags, clocks, clip
            # ids and reset start times still must not move at all)
            "bf16+bf16_storage": dict(obs=5e-4, action=8e-4, logp_r=1e-4, logp_a=1e-3, reward_r=2e-2, reward_a=3e-2, tar_r=2e-2, tar_a=1.2e-1)}
# measured on MI355X (max abs error over the subset): bf16x3 obs 2.4e-7, action 1.2e-7, reward 1.7e-6, tar_val 1.1e-5;
# bf16 obs 4.4e-6, action 4.8e-6, reward 9.2e-5, tar_val 5.2e-4; bf16 + bf16_storage rollout obs 2.2e-4, action 3.1e-4, reward 1.2e-2, tar_val 5.5e-2


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "bf16+bf16_storage"])
def test_16384_envs_five_clips_subset_matches_oracle(precision):
    """BASELINE configs[2] composition: 16 384 envs, a multi-clip library (reference-compatible raw-frame offsets), the bf16 MFMA MLP
    path -- agent.matmul_precision=bf16x3 (exact split) and =bf16 (the bf16-STORAGE mode the configs[2] throughput is quoted for) --
    one rollout + build-train-data; envs 0, 64, 128, ... (256 of them) against the oracle (restating base_agent.py:427-470,
    ppo_agent.py:111-159, add_agent.py:110-139)."""
    import torch
    import add_gym_amd.learning.add_agent as A
    from oracle import loop as LP
    from oracle import task as OT
    from oracle.motion import MotionLib as OracleLib
    from tests.util import oracle_kin
    from add_gym_amd.anim.synth import synth_clip

    N, Tn, C, NF = 16384, 32, 5, 60  # 2 s clips: a good share of the envs runs past a clip end within the rollout
    tol_key = precision
    precision, _, roll = precision.partition("+")
    cfg = make_cfg(N, steps_per_iter=Tn, matmul_precision=precision, **({"rollout_precision": roll} if roll else {}))
    cfg["task"]["motion_file"] = f"synthetic:{C}x{NF}"
    ag = A.ADDAgent(cfg)
    assert ag._storage16 == (1 if precision == "bf16" else 0) and ag._roll_storage == bool(roll)
    tol = CFG2_TOL[tol_key]
    seed = 21
    params = OL.synth_params(seed)
    ag._model.load({k: torch.tensor(v) for k, v in params.items()})
    lib_p = ag._motion_lib
    order = list(cfg["task"]["motion_joint_order"])
    kin_p = ag._env.robot._kin_char_model
    frames = [synth_clip(kin_p, order, c, NF) for c in range(C)]
    lib = OracleLib(frames, [1.0] * C, order, oracle_kin(), 0.01, True)
    # identical table rows on both sides (the product's host ingest is pinned bit for bit in tests/test_host_ingest.py)
    hp, hv = lib_p.host_pose.numpy(), lib_p.host_vel.numpy()
    lib.step = dict(root_pos=hp[:, 0:3].copy(), root_rot=hp[:, 3:7].copy(), dof_pos=hp[:, 7:36].copy(), root_vel=hv[:, 0:3].copy(),
                    root_ang_vel=hv[:, 3:6].copy(), dof_vel=hv[:, 6:35].copy())
    assert lib.total_steps == lib_p.total_steps

    rng = np.random.RandomState(4)
    sub = np.arange(0, N, 64)
    n = len(sub)
    # draws for every env and step: clip, segment, jitter (consumed only where a reset happens)
    clip = rng.randint(0, C, (Tn + 1, N))
    seg = rng.randint(0, 20, (Tn + 1, N))
    jit = rng.rand(Tn + 1, N).astype(F)
    noise = rng.standard_normal((Tn, N, 29)).astype(F)
    cdf = np.cumsum(np.full(C, 1.0 / C, F), dtype=F)
    u_clip = np.where(clip == 0, 0.5 * cdf[0], 0.5 * (cdf[np.maximum(clip - 1, 0)] + cdf[clip])).astype(F)
    u_seg = ((seg + 0.5) / 20.0).astype(F)  # sampler errors are all ones during the first iteration: uniform segments

    def uni(k):
        return T(np.stack([u_clip[k], u_seg[k], jit[k]]))

    inj = {ag.stream_reset_all(0): uni(0)}
    for t in range(Tn):
        inj[ag.stream_train_reset(t)] = uni(t + 1)
    ag.inject = dict(noise=T(noise), uniforms=inj)
    ag.reset_all_envs()
    ag._init_train()
    ag._rollout_train()
    ag._build_train_data()
    torch.cuda.synchronize()

    orc = LP.Agent(LP.AgentCfg(steps_per_iter=Tn), OT.TaskCfg(), lib, n, params)
    orc.init(dict(ids=clip[0][sub], segments=seg[0][sub], jitter=jit[0][sub]))
    draws = LP.Draws(noise[:, sub], lambda t, ids: dict(ids=clip[t + 1][sub[ids]], segments=seg[t + 1][sub[ids]], jitter=jit[t + 1][sub[ids]]), None)
    orc.rollout(draws)
    ob = orc.buf
    task_reward = ob["reward"].copy()
    orc.build_train_data()
    B = ag._B
    s = torch.tensor(sub, device="cuda")
    done = B["done"][:, s].cpu().numpy()
    assert np.array_equal(done, ob["done"])  # bit-exact flags over the whole rollout of the subset
    assert (done != 0).sum() > 20 and len(np.unique(done)) >= 2
    assert np.array_equal(B["motion_time"][:, s].cpu().numpy(), ob["motion_times"])  # bit-exact clocks / reset start times
    assert np.array_equal(B["motion_id"][:, s].cpu().numpy(), ob["motion_ids"])
    obs = B["obs"][:, s].cpu().numpy()
    got = dict(obs=obs[:Tn, :, :264], disc_obs=B["disc_obs"][:Tn, s].cpu().numpy()[..., :114], disc_demo=B["disc_demo"][:Tn, s].cpu().numpy()[..., :114],
               action=B["action"][:Tn, s].cpu().numpy()[..., :29], a_logp=B["a_logp"][:Tn, s].cpu().numpy(), reward=B["reward"][:, s].cpu().numpy(),
               tar_val=B["tar_val"][:, s].cpu().numpy())
    want = dict(obs=ob["obs"], disc_obs=ob["disc_obs"], disc_demo=ob["disc_obs_demo"], action=ob["action"], a_logp=ob["a_logp"], reward=ob["reward"],
                tar_val=ob["tar_val"])
    print("configs[2] %s max abs error vs oracle: " % precision + ", ".join("%s %.2e" % (k, np.abs(got[k] - want[k]).max()) for k in got))
    np.testing.assert_allclose(got["obs"], want["obs"], rtol=0, atol=tol["obs"])
    # next_obs[t] is obs[t+1] wherever the env was not reset at step t
    keep = done == 0
    np.testing.assert_allclose(obs[1:, :, :264][keep], ob["next_obs"][keep], rtol=0, atol=tol["obs"])
    np.testing.assert_allclose(got["disc_obs"], want["disc_obs"], rtol=0, atol=tol["obs"])
    np.testing.assert_allclose(got["disc_demo"], want["disc_demo"], rtol=0, atol=5e-5)  # reference rows only: independent of the MLPs
    np.testing.assert_allclose(got["action"], want["action"], rtol=0, atol=tol["action"])
    np.testing.assert_allclose(got["a_logp"], want["a_logp"], rtol=tol["logp_r"], atol=tol["logp_a"])
    # after build-train-data the reward buffer holds the discriminator reward (task_reward_weight 0), tar_val the TD(lambda) target
    np.testing.assert_allclose(got["reward"], want["reward"], rtol=tol["reward_r"], atol=tol["reward_a"])
    np.testing.assert_allclose(got["tar_val"], want["tar_val"], rtol=tol["tar_r"], atol=tol["tar_a"])
    assert np.isfinite(task_reward).all()


def test_full_library_shard_43_clips_4096_envs():
    """BASELINE configs[3], one GPU's shard: 4096 envs on a 43-clip library (the size of add-gym's assets/motions), one whole
    training iteration; every clip is in use, the sampler's per-clip error table is updated, nothing leaves the tables."""
    import torch
    import add_gym_amd.learning.add_agent as A

    cfg = make_cfg(4096, steps_per_iter=32)
    cfg["task"]["motion_file"] = "synthetic:43x300"
    ag = A.ADDAgent(cfg)
    lib = ag._motion_lib
    assert lib.get_num_motions() == 43 and lib.total_steps == 43 * 997
    ag.reset_all_envs()
    ag._init_train()
    info = ag._train_iter()
    torch.cuda.synchronize()
    assert all(np.isfinite(v) for v in info.values())
    ids = ag._B["motion_id"].cpu().numpy()
    assert set(np.unique(ids)) == set(range(43))
    err = ag._smp["errors"].cpu().numpy()
    assert err.shape == (43, 20) and (err != 1.0).sum() > 400 and np.isfinite(err).all()  # EMA moved in the (clip, segment) cells that were visited
    assert torch.isfinite(ag._B["obs"]).all() and torch.isfinite(ag._B["disc_demo"]).all()
    # reference-compatible indexing (raw-frame clip offsets, motion_lib.py:280-282, 322-326): every gathered row index stays inside the tables
    t = ag._S["time"] + ag._S["time_off"]
    idx = lib.step_index(ag._S["motion_id"], t).cpu().numpy()
    assert idx.min() >= 0 and idx.max() < lib.total_steps


def test_randomised_rigid_shard_8192_envs_inside_the_graph_rollout():
    """BASELINE configs[4] as written, one GPU's shard (65 536 envs over 8 GPUs = 8192 per GPU): the rigid-body engine WITH domain
    randomisation (a build-defined extension: add-gym has none) inside the hipGraph-captured rollout.  The per-env gain / friction
    redraws and the root pushes are drawn on the device from a device-resident step counter (addhip_rigid_randomize), so the rollout
    replayed from the graph must be BIT-IDENTICAL to the same rollout issued call by call -- through the eager warm-up, the three
    captures (one per ring phase) and pure replays, with redraws (every 5 control steps) and pushes (every 7) falling inside
    replays; and a gain change after the captures must reach the replays (the graphs are re-captured, the tables refreshed in place)."""
    import torch
    import add_gym_amd.learning.add_agent as A
    from add_gym_amd.config import load_config
    from tests.util import kin_meta

    keys = ("obs", "action", "a_logp", "done", "reward", "motion_time", "disc_obs")
    runs = []
    for graph in (False, True):
        cfg = load_config("train", ["engine=rigid", "engine.num_envs=8192", "agent.steps_per_iter=16", "task.motion_file=synthetic:3x120", "seed=11",
                                    "engine.domain_randomization.enabled=true", "engine.domain_randomization.push_interval=7",
                                    "engine.domain_randomization.resample_interval=5", "engine.domain_randomization.seed=5",
                                    f"agent.rollout_graph={str(graph).lower()}"])
        cfg["task"]["motion_joint_order"] = kin_meta()["motion_joint_order"]
        ag = A.ADDAgent(cfg)
        ent = ag._env.robot.entity
        assert ent.env_scale is not None and ag._graph_ok()
        ag.reset_all_envs()
        ag._init_train()
        out = []
        for it in range(8):
            if it == 6:  # a model change after the captures: stiffer knees from here on
                kp = ent._kp.clone()
                kp[6:] *= 1.5
                ent.set_dofs_kp(kp)
                if graph:
                    ptrs = (ent._d_body.data_ptr(), ent._d_chains.data_ptr())
            if it:
                ag._B["obs"][0].copy_(ag._B["obs"][ag.T])
            ag._rollout_train()
            ag._iter += 1
            torch.cuda.synchronize()
            out.append(dict({k: ag._B[k].clone() for k in keys}, pose=ag._S["sim_pose"].clone(), vel=ag._S["sim_vel"].clone(), scale=ent.env_scale.clone(),
                            steps=int(ent._d_steps[0]), ticket=int(ent._d_steps[1])))
            if graph and it == 5:
                assert len(ag._graphs) == 3  # T = 16: the ring phase advances by 1 per iteration
        runs.append(out)
        if graph:
            assert 1 <= len(ag._graphs) <= 2 and ptrs == (ent._d_body.data_ptr(), ent._d_chains.data_ptr())  # re-captured; tables refreshed in place
            info = ag._train_iter()  # and a whole iteration (rollout from the graph + update) stays finite
            torch.cuda.synchronize()
            assert all(np.isfinite(v) for v in info.values())
    for it, (a, b) in enumerate(zip(*runs)):
        assert a["steps"] == b["steps"] == 16 * (it + 1) and a["ticket"] == b["ticket"] == 0
        for k in a:
            if k not in ("steps", "ticket"):
                assert torch.equal(a[k], b[k]), (it, k)
    eager = runs[0]
    assert not torch.equal(eager[0]["scale"], eager[1]["scale"])          # redrawn (steps 20, 25, 30 fall in iteration 1)
    assert 0.8 <= float(eager[-1]["scale"][:, 0].min()) and float(eager[-1]["scale"][:, 0].max()) <= 1.2
    assert int((eager[-1]["done"] == 1).sum()) > 0 and torch.isfinite(eager[-1]["pose"]).all()


def test_bf16_storage_mode_tracks_fp32_training():
    """agent.matmul_precision=bf16 (bf16 activations / gradients / weight shadow in HBM for the update step, fp32 master weights and
    accumulation): six iterations from the same seeds stay close to the fp32-MFMA run -- losses and discriminator statistics within
    10 %, identical rollouts at iteration 0 (the rollout runs on fp32 operands), parameters within the distance Adam travels."""
    import torch
    import add_gym_amd.learning.add_agent as A

    runs = {}
    for prec in ("fp32", "bf16"):
        cfg = make_cfg(1024, steps_per_iter=32, matmul_precision=prec)
        cfg["task"]["motion_file"] = "synthetic:2x600"
        cfg["seed"] = 3
        ag = A.ADDAgent(cfg)
        assert ag._storage16 == (1 if prec == "bf16" else 0)
        ag.reset_all_envs()
        ag._init_train()
        rows = []
        for it in range(6):
            info = ag._train_iter()
            ag._iter += 1
            rows.append(info)
            if it == 0:
                first_done = ag._B["done"].clone()
        torch.cuda.synchronize()
        runs[prec] = (rows, first_done, ag._model.export())
        del ag
    a, b = runs["fp32"], runs["bf16"]
    assert torch.equal(a[1], b[1])  # same weights, same draws: the first rollout's done flags coincide
    for it in range(6):
        for k in ("critic_loss", "disc_loss", "disc_grad_penalty", "disc_reward_mean", "adv_std", "mean_return"):
            x, y = float(a[0][it][k]), float(b[0][it][k])
            assert np.isfinite(y) and abs(x - y) <= 0.10 * max(abs(x), abs(y)) + 2e-3, (it, k, x, y)
    # 240 optimiser steps at lr 1e-4: nobody moved further than 0.024 from the start, so two runs are at most 0.048 apart.  (The worst element
    # -- a discriminator bias whose gradient sign is noise -- was seen between 2.9e-2 and 3.5e-2 over repeated runs of the same build: the
    # default mode's float atomics make the 240-step trajectory differ from run to run.)
    for k, v in a[2].items():
        assert float((v - b[2][k]).abs().max()) <= 4e-2, k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_deterministic_update_step_repeats_bit_for_bit(precision):
    """agent.deterministic: the 16 384-row update step (all three nets on their streams) run twice from the same buffers gives bit-identical
    gradients -- all 22 tensors, padding included -- where the default (float atomics behind bias gradients and loss-head sums) differs in
    the last bits from run to run; and the deterministic gradients are the default ones to fp32 accuracy."""
    import torch
    import add_gym_amd.learning.add_agent as A

    grads = {}
    for det in (True, False):
        cfg = make_cfg(4096, steps_per_iter=32, matmul_precision=precision)
        cfg["task"]["motion_file"] = "synthetic:1x300"
        cfg["agent"]["deterministic"] = det
        ag = A.ADDAgent(cfg)
        assert ag._deterministic == det and ag.Mb == 16384
        params = OL.synth_params(11)
        ag._model.load({k: torch.tensor(v) for k, v in params.items()})
        _fill_minibatch(ag, OL.Model(params), 5)
        runs = []
        for _ in range(3 if det else 1):
            ag._W["stats"].zero_()
            ag._run_update_sections()
            torch.cuda.synchronize()
            runs.append(ag._model.grads.clone())
        grads[det] = runs
        del ag
    assert torch.equal(grads[True][0], grads[True][1]) and torch.equal(grads[True][0], grads[True][2])
    assert float(grads[True][0].abs().max()) > 0.0
    d = (grads[True][0] - grads[False][0]).abs().max() / grads[False][0].abs().max()
    assert float(d) < (2e-2 if precision == "bf16" else 1e-4), float(d)

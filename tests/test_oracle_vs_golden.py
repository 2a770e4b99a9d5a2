"""Pins the CPU oracle (oracle/) to the golden vectors generated from the reference's own
Python (tools/gen_golden.py).  Runs on CPU; no GPU, no /root/reference needed."""
import json

import numpy as np
import pytest

from oracle import learn as L
from oracle import loop as LP
from oracle import quat as Q
from oracle import task as T
from oracle.motion import MotionLib, torch_cpu_arange
from tests.util import gload, kin_meta, oracle_kin, oracle_lib, variant

F = np.float32
TOL = dict(rtol=2e-6, atol=2e-6)  # fp32: a few ulp of the transcendental chains (numpy vs torch libm)


def close(a, b, **kw):
    tol = {**TOL, **kw}
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), **tol)


def test_quat_math():
    g = gload("quat_math")
    q0, q1, v, t = g["q0"], g["q1"], g["v"], g["t"]
    close(Q.quat_mul(q0, q1), g["quat_mul"])
    close(Q.quat_rotate(q0, v), g["quat_rotate"])
    close(Q.quat_pos(q0), g["quat_pos"], rtol=0, atol=0)
    close(Q.quat_conjugate(q0), g["quat_conjugate"], rtol=0, atol=0)
    ax, an = Q.quat_to_axis_angle(q0)
    close(ax, g["axis_angle_axis"])
    close(an, g["axis_angle_angle"])
    close(Q.quat_to_exp_map(q0), g["quat_to_exp_map"])
    close(Q.quat_to_tan_norm(q0), g["quat_to_tan_norm"])
    close(Q.quat_diff_angle(q0, q1), g["quat_diff_angle"], atol=2e-6)
    close(Q.slerp(q0, q1, t), g["slerp"])
    close(Q.calc_heading(q0), g["calc_heading"])
    close(Q.calc_heading_quat_inv(q0), g["calc_heading_quat_inv"])
    close(Q.axis_angle_to_quat(g["axis"], g["angle"]), g["axis_angle_to_quat"])
    close(Q.quat_twist_angle(q0, g["unit_axis"]), g["quat_twist_angle"])
    close(Q.quat_normalize(q0 * F(1.7)), g["quat_normalize"])


def test_kin_tree():
    g = gload("kin_tree")
    meta = kin_meta()
    kin = oracle_kin()
    assert kin.body_names == meta["body_names"]
    assert kin.joint_names == meta["joint_names"]
    assert np.array_equal(np.asarray(kin.parents), g["parents"])
    close(kin.axes, g["axes"][1:], rtol=0, atol=0)
    lo, hi, a_mean, a_std = kin.action_bounds()
    close(lo, g["action_space"][:, 0], rtol=0, atol=0)
    close(hi, g["action_space"][:, 1], rtol=0, atol=0)
    order = meta["motion_joint_order"]
    assert [order.index(n) for n in kin.joint_names[1:]] == list(g["motion_idx"])


def _slerp_ambiguous(lib):
    """(row, dof) cells whose raw-frame pair sits on the reference's slerp fallback threshold
    |sin(half angle)| < 1e-3 (torch_util.py:320): there a 1-ulp difference in sin/cos decides
    between slerp and the plain average, so libm differences flip the branch."""
    amb = []
    for m in range(lib.num_motions()):
        n = int(lib.step_count[m])
        ids = np.full(n, m)
        times = torch_cpu_arange(n, lib.dt)
        nf1 = F(lib.num_frames[m] - 1)
        ph = lib.calc_phase(ids, times)
        i0 = (ph * nf1).astype(np.int64)
        i1 = np.minimum(i0 + 1, lib.num_frames[m] - 1)
        j0 = lib.frame["joint_rot"][i0 + lib.frame_start[m]]
        j1 = lib.frame["joint_rot"][i1 + lib.frame_start[m]]
        c = np.abs(np.sum(j0 * j1, axis=-1, dtype=F))
        s = np.sqrt(np.maximum(F(1) - c * c, 0))
        # ... and the second fallback |cos| >= 1 -> q0 (torch_util.py:321) is decided by the last ulp of cos
        amb.append((np.abs(s - 1e-3) < 2.5e-4) | (c >= 1 - 2e-7))
    return np.concatenate(amb, axis=0)


@pytest.mark.parametrize("two", [False, True])
def test_motion_tables(two):
    g = gload("motion_small")
    lib = oracle_lib(two)
    pre = "two_" if two else ""
    for n in ("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_vel"):
        assert lib.step[n].shape == g[pre + "step_" + n].shape
        close(lib.step[n], g[pre + "step_" + n], atol=1e-5)
    amb = _slerp_ambiguous(lib)
    d = np.abs(lib.step["dof_pos"] - g[pre + "step_dof_pos"])
    assert amb.mean() < 0.5  # slow joints (sub-milliradian per raw frame) all sit on the fallbacks
    assert d[~amb].max() < 2e-6
    assert d.max() < 2e-3  # ambiguous cells differ by at most (0.5-blend)*|q1-q0|
    close(lib.lengths, g[pre + "lengths"], rtol=0, atol=0)
    assert np.array_equal(lib.frame_start, g[pre + "start_idx"])
    close(lib.weights, g[pre + "weights"], rtol=0, atol=0)


def test_lookup_indices_bit_exact():
    g = gload("lookup")
    lib = oracle_lib()
    # fp32 accumulated clock (env.py:155) reproduced exactly
    c, acc = F(0), []
    for _ in range(2000):
        c = F(c + F(0.01))
        acc.append(c)
    assert np.array_equal(np.asarray(acc, F), g["acc_clock"])
    assert np.array_equal(lib.step_index(g["ids"], g["times"]), g["idx"])
    close(lib.get_step(g["ids"], g["times"])[0], g["root_pos"], atol=2e-6)
    lib2 = oracle_lib(two=True)
    assert np.array_equal(lib2.step_index(g["two_ids"], g["two_times"]), g["two_idx"])
    # the multi-clip start-offset quirk is what makes these differ:
    fixed = oracle_lib(two=True, reference_compat=False)
    assert not np.array_equal(fixed.step_index(g["two_ids"], g["two_times"]), g["two_idx"])


VARIANTS = {
    "default": {},
    "local": dict(global_obs=False),
    "noheight": dict(root_height_obs=False),
    "local_noheight": dict(global_obs=False, root_height_obs=False),
    "vel_phase": dict(enable_vel_obs=True, enable_phase_obs=True),
    "local_vel": dict(global_obs=False, enable_vel_obs=True),
    # task.num_disc_obs_steps = 2 (fixture obs_reward_done_s2): a two-deep history ring, two clip frames per demo observation
    "two_steps": dict(num_disc_obs_steps=2),
    "two_steps_local_vel": dict(num_disc_obs_steps=2, global_obs=False, enable_vel_obs=True),
    # = 4 (fixture obs_reward_done_s4)
    "four_steps": dict(num_disc_obs_steps=4),
    "four_steps_local_vel": dict(num_disc_obs_steps=4, global_obs=False, enable_vel_obs=True),
}
FIELDS = T.TaskState.FIELDS


def step_fixture(vname):
    """The fixture that holds a variant of the env-step case."""
    return ("obs_reward_done_jw" if vname == "joint_w" else "obs_reward_done_s2" if vname.startswith("two_steps") else
            "obs_reward_done_s4" if vname.startswith("four_steps") else "obs_reward_done")


def _task_from_fixture(v, cfg, lib, n, prefix="", sim_prefix=None):
    ts = T.TaskState(cfg, lib, n)
    ts.time = v[prefix + "time"].astype(F).copy()
    ts.time_off = v[prefix + "time_off"].astype(F).copy()
    ts.motion_ids = v[prefix + "motion_ids"].astype(np.int64).copy()
    for k in FIELDS:
        ts.hist[k] = v[prefix + "hist_" + k].copy()
    ts.head = int(v[prefix + "hist_head"])
    return ts


@pytest.mark.parametrize("vname", list(VARIANTS))
def test_obs_reward_done(vname):
    v = variant(gload(step_fixture(vname)), vname)
    cfg = T.TaskCfg(**VARIANTS[vname])
    lib = oracle_lib(golden_tables=True)
    n = v["time"].shape[0]
    ts = _task_from_fixture(v, cfg, lib, n)
    sim = tuple(v[k] for k in FIELDS)
    obs, d_obs, d_demo, r, done = ts.step(sim, v["contact"])
    assert np.array_equal(ts.time, v["time_post"])
    for k, ref in zip(FIELDS, ts.ref):
        close(ref, v["ref_" + k], rtol=0, atol=0)
    close(obs, v["obs"], atol=3e-6)
    close(d_obs, v["disc_obs"], atol=3e-6)
    close(d_demo, v["disc_obs_demo"], atol=3e-6)
    close(r, v["reward"], atol=3e-6)
    assert done.dtype == np.int32 and np.array_equal(done, v["done"])
    if vname == "two_steps":
        assert d_obs.shape[1] == 76 and ts.hist["root_pos"].shape[1] == 2
    if vname == "four_steps":
        assert d_obs.shape[1] == 152 and ts.hist["root_pos"].shape[1] == 4
    if vname == "default":
        assert obs.shape[1] == 264 and d_obs.shape[1] == 114
        assert set(np.unique(done)) == {0, 1, 2, 3}  # every flag occurs


def test_obs_reward_done_joint_error_weights():
    """task.joint_err_w (add_reward.py:24-52): weighted pose / velocity error sums in the reward; the done flags keep the
    unweighted mean (add_done.py:129-132)."""
    v = variant(gload("obs_reward_done_jw"), "joint_w")
    cfg = T.TaskCfg(dof_err_w=v["dof_err_w"])
    lib = oracle_lib(golden_tables=True)
    n = v["time"].shape[0]
    ts = _task_from_fixture(v, cfg, lib, n)
    obs, d_obs, d_demo, r, done = ts.step(tuple(v[k] for k in FIELDS), v["contact"])
    close(obs, v["obs"], atol=3e-6)
    close(r, v["reward"], atol=3e-6)
    assert np.array_equal(done, v["done"])
    base = variant(gload("obs_reward_done"), "default")
    assert np.array_equal(v["done"], base["done"]) and np.abs(v["reward"] - base["reward"]).max() > 1e-3  # same states, other reward


@pytest.mark.parametrize("tag", ["one", "two"])
def test_reset(tag):
    _reset_case("reset", tag, T.TaskCfg())


@pytest.mark.parametrize("steps", [2, 4])
def test_reset_other_disc_obs_steps(steps):
    """task.num_disc_obs_steps = 2 / 4: CircularBuffer.fill writes the clip frames t-(S-1)dt .. t (circular_buffer.py:22-29)."""
    _reset_case(f"reset_s{steps}", "one", T.TaskCfg(num_disc_obs_steps=steps))


def _reset_case(fixture, tag, cfg):
    v = variant(gload(fixture), tag)
    lib = oracle_lib(two=(tag == "two"), golden_tables=True)
    n = v["time"].shape[0]
    ts = _task_from_fixture(v, cfg, lib, n)
    ts.sampler.errors = v["sampler_errors"].copy()
    env_ids = v["env_ids"]
    close(ts.sampler.probs(v["draw_ids"]), v["probs"], atol=1e-7)
    qpos, qvel = ts.reset(env_ids, v["draw_ids"], v["draw_segments"], v["draw_jitter"])
    assert np.array_equal(ts.time, v["post_time"])
    assert np.array_equal(ts.time_off, v["post_time_off"])  # start times bit-exact (quantise + clamp)
    assert np.array_equal(ts.motion_ids, v["post_motion_ids"])
    sim = {k: v["sim_" + k].copy() for k in FIELDS}
    sim["root_pos"][env_ids], sim["root_rot"][env_ids], sim["dof_pos"][env_ids] = qpos[:, :3], qpos[:, 3:7], qpos[:, 7:]
    sim["root_vel"][env_ids], sim["root_ang_vel"][env_ids], sim["dof_vel"][env_ids] = qvel[:, :3], qvel[:, 3:6], qvel[:, 6:]
    for k in FIELDS:
        close(sim[k], v["post_sim_" + k], rtol=0, atol=0)
        close(ts.hist[k], v["post_hist_" + k], rtol=0, atol=0)
    obs, d_obs, d_demo = ts.compute_obs(tuple(sim[k] for k in FIELDS))
    close(obs, v["obs"], atol=3e-6)
    close(d_obs, v["disc_obs"], atol=3e-6)
    close(d_demo, v["disc_obs_demo"], atol=3e-6)


def test_sampler():
    g = gload("sampler")
    s = T.SegmentSampler(g["lengths"], 0.01, 20, None, 0.02)
    close(s.segment_sizes, g["segment_sizes"], rtol=0, atol=0)
    s.update_errors(g["ids"], g["times"], g["err"])
    close(s.errors, g["errors1"], rtol=1e-5)
    s.update_errors(g["ids2"], g["times2"], g["err2"])
    close(s.errors, g["errors2"], rtol=1e-5)
    s.errors = g["errors2"].copy()
    close(s.probs(g["probs_ids"]), g["probs"], atol=1e-7)
    close(s.probs(np.arange(3)), g["probs_all"], atol=1e-7)
    fd = (T.torch_floor_divide_f32(g["fd_in"], 0.01) * F(0.01)).astype(F)
    assert np.array_equal(fd, g["fd_out"])


def test_normalizers():
    g = gload("normalizers")
    nm, dn = L.Normalizer(7), L.DiffNormalizer(5)
    for it in range(3):
        for s in range(4):
            nm.record(g[f"x{it}_{s}"])
            dn.record(g[f"y{it}_{s}"])
        nm.update()
        dn.update()
        close(nm.mean, g[f"mean{it}"], rtol=1e-5)
        close(nm.std, g[f"std{it}"], rtol=1e-5)
        assert nm.count == int(g[f"count{it}"][0]) and dn.count == int(g[f"dcount{it}"][0])
        close(dn.mean_abs, g[f"mean_abs{it}"], rtol=1e-5)
    nm.mean, nm.std, dn.mean_abs = g["mean2"], g["std2"], g["mean_abs2"]
    close(nm.normalize(g["xq"]), g["xq_norm"])
    close(nm.unnormalize(g["xq"]), g["xq_unnorm"])
    close(dn.normalize(g["yq"]), g["yq_norm"])


def golden_logstd(g_or_name):
    """synth_params' logstd argument for a fixture: False (FIXED), True (CONSTANT: a vector), "variable" (VARIABLE: a second head)."""
    if isinstance(g_or_name, str):
        return "variable" if g_or_name.endswith("variable_std") else g_or_name.endswith("constant_std")
    return False if "logstd" not in g_or_name.files else ("variable" if int(g_or_name["logstd"]) == 2 else True)


@pytest.mark.parametrize("name", ["actor_step", "actor_step_constant_std", "actor_step_variable_std"])
def test_actor_step(name):
    """(`_constant_std`: actor_std_type CONSTANT -- a standard deviation per action dimension, distribution_gaussian_diag.py:32-37, 47-58;
    `_variable_std`: VARIABLE -- per sample and dimension, from a second head, :38-43, 52-53.)"""
    g = gload(name)
    model = L.Model(L.synth_params(int(g["seed"]), logstd=golden_logstd(name)))
    on = L.Normalizer(264, g["obs_mean"], g["obs_std"])
    an = L.Normalizer(29, g["a_mean"], g["a_std"])
    a, logp, _ = L.actor_step(model, on, an, g["obs"], g["noise"], g["rand_action_mask"])
    close(a, g["action"], atol=1e-5)
    close(logp, g["a_logp"], rtol=1e-5, atol=1e-4)


def test_td_lambda_adv():
    g = gload("td_lambda_adv")
    ret = L.td_lambda_return(g["r"], g["next_vals"], g["done"], 0.99, 0.95)
    close(ret, g["ret_raw"], rtol=1e-6, atol=1e-6)
    nv = g["next_vals"].copy()
    nv[(g["done"] == 1) | (g["done"] == 2)] = 0
    ret2 = L.td_lambda_return(g["r"], nv, g["done"], 0.99, 0.95)
    close(ret2, g["tar_val"], rtol=1e-6, atol=1e-6)
    adv, mean, std = L.advantages(ret2, g["vals"], np.ones_like(ret2), 4.0)
    close(mean, g["adv_mean"], rtol=1e-5)
    close(std, g["adv_std"], rtol=1e-5)
    close(adv, g["adv"], rtol=1e-5, atol=1e-5)


def _summary(x):
    f = np.asarray(x, np.float64).reshape(-1)
    stride = max(1, f.size // 64)
    return f.sum(), np.sqrt(np.square(f).sum()), np.asarray(x, F).reshape(-1)[::stride][:64]


def _check_summary(g, prefix, named, rtol, atol_scale=1.0):
    for name, val in named.items():
        s, l2, sample = _summary(val)
        ref_l2 = float(g[f"{prefix}.{name}.l2"])
        assert abs(l2 - ref_l2) <= rtol * max(ref_l2, 1e-12), (prefix, name, l2, ref_l2)
        assert abs(s - float(g[f"{prefix}.{name}.sum"])) <= rtol * ref_l2 * np.sqrt(np.asarray(val).size) + 1e-12, (prefix, name)
        np.testing.assert_allclose(sample, g[f"{prefix}.{name}.sample"], rtol=rtol * 20, atol=rtol * atol_scale * max(ref_l2 / np.sqrt(np.asarray(val).size), 1e-12) * 20)


def golden_nets(g):
    """The net modules a losses fixture was generated with (None: add_g1.yaml's)."""
    import json

    return json.loads(str(g["nets"])) if "nets" in g.files else None


def golden_loss_cfg(g):
    """LossCfg with the agent options a losses fixture was generated with."""
    import json

    return L.LossCfg(**(json.loads(str(g["agent_over"])) if "agent_over" in g.files else {}))


@pytest.mark.parametrize("name", ["losses", "losses_small_nets", "losses_constant_std", "losses_disc3", "losses_constant_std_entropy", "losses_variable_std", "losses_variable_std_entropy"])
def test_losses_grads_adamw(name):
    g = gload(name)
    model = L.Model(L.synth_params(int(g["seed"]), nets=golden_nets(g), logstd=golden_logstd(g)))
    on = L.Normalizer(264, g["obs_mean"], g["obs_std"])
    an = L.Normalizer(29, g["a_mean"], g["a_std"])
    dn = L.DiffNormalizer(114)
    dn.mean_abs = g["disc_mean_abs"]
    mb = dict(norm_obs=on.normalize(g["in.obs"]), norm_action=an.normalize(g["in.action"]), a_logp=g["in.a_logp"],
              adv=g["in.adv"], tar_val=g["in.tar_val"], rand_action_mask=g["in.rand_action_mask"],
              norm_diff=dn.normalize(g["in.disc_obs_demo"] - g["in.disc_obs"]))
    opt = L.AdamW(model, 1e-4)
    for step in range(3):
        loss, info = L.compute_loss(model, golden_loss_cfg(g), mb)
        grads = opt.step(loss)
        if step == 0:
            if "info.action_entropy" in g.files:
                np.testing.assert_allclose(info["action_entropy"], float(g["info.action_entropy"]), rtol=1e-6)
            for k in ("loss", "actor_loss", "critic_loss", "disc_loss", "clip_frac", "imp_ratio", "action_bound_loss",
                      "disc_grad_penalty", "disc_logit_loss", "disc_pos_acc", "disc_neg_acc", "disc_pos_logit", "disc_neg_logit"):
                # |logp| ~ 2e3 in this fixture (12-sigma actions) -> fp32 ratio noise ~1e-4 absolute on the actor terms
                np.testing.assert_allclose(info[k], float(g["info." + k]), rtol=2e-5, atol=1e-4, err_msg=k)
            _check_summary(g, "grad", grads, rtol=2e-4)
        if step in (0, 2):
            _check_summary(g, f"param{step + 1}", {k: v.detach().numpy() for k, v in model.p.items()}, rtol=1e-5)


LOOP_VARIANTS = {"loop_1iter": dict(two=False, task={}), "loop_1iter_two": dict(two=True, task={}),
                 "loop_1iter_time": dict(two=False, task=dict(max_episode_length=0.4)),
                 "loop_1iter_s2": dict(two=False, task=dict(num_disc_obs_steps=2)),
                 "loop_1iter_s4": dict(two=False, task=dict(num_disc_obs_steps=4))}


@pytest.mark.parametrize("name", list(LOOP_VARIANTS))
def test_loop_one_iteration(name):
    """One whole reference iteration replayed: single clip; two clips (clip draws, raw-frame table offsets, [2,20] sampler table);
    0.4 s episode limit with pre-aged episode clocks (DONE_TIME samples bootstrap from the pre-reset next_obs)."""
    g = gload(name)
    v = LOOP_VARIANTS[name]
    n = g["noise"].shape[1]
    lib = oracle_lib(two=v["two"], golden_tables=True)
    ag = LP.Agent(LP.AgentCfg(), T.TaskCfg(**v["task"]), lib, n, L.synth_params(int(g["seed"]), disc_dim=38 * v["task"].get("num_disc_obs_steps", 3)))
    ag.init(dict(ids=g["init_ids"], segments=g["init_segments"], jitter=g["init_jitter"]))
    ag.task.time = (ag.task.time + g["time_preset"]).astype(np.float32)          # (zeros except in the time variant)
    ag.task.time_off = (ag.task.time_off - g["time_preset"]).astype(np.float32)
    Tn = g["noise"].shape[0]
    resets = []
    for t in range(Tn):
        k = int(g["reset_count"][t])
        resets.append(dict(ids=g["reset_ids"][t, :k], segments=g["reset_segments"][t, :k], jitter=g["reset_jitter"][t, :k]))
    draws = LP.Draws(g["noise"], resets, g["perms"])
    plan = g["contact_plan"]
    contact = [(plan[t] >= 0) for t in range(Tn)]  # link 5 is a non-foot body -> contact flag
    info = ag.train_iter(draws, contact)
    b = ag.buf
    assert np.array_equal(b["done"], g["buf.done"])  # bit-exact flags over the whole rollout
    assert np.array_equal(b["motion_times"], g["buf.motion_times"])  # bit-exact clocks / reset times
    for t in range(Tn):
        assert int(np.sum(b["done"][t] != 0)) == int(g["reset_count"][t])
    close(b["obs"][-1], g["buf.obs_last"], atol=2e-5)
    close(b["reward"], g["buf.reward"], rtol=2e-4, atol=2e-5)
    close(b["adv"], g["buf.adv"], rtol=2e-3, atol=2e-3)
    for k in ("obs", "next_obs", "action", "a_logp", "disc_obs", "disc_obs_demo", "tar_val"):
        ref = float(g[f"buf.{k}.abs"])
        assert abs(np.abs(b[k].astype(np.float64)).sum() - ref) <= 2e-5 * ref, k
    for k in ("adv_mean", "adv_std", "disc_reward_mean", "disc_reward_std", "loss", "actor_loss", "critic_loss", "disc_loss",
              "clip_frac", "imp_ratio", "disc_grad_penalty", "disc_logit_loss", "disc_pos_acc", "disc_neg_acc",
              "disc_pos_logit", "disc_neg_logit", "mean_return", "mean_ep_len", "num_eps"):
        np.testing.assert_allclose(info[k], float(g["info." + k]), rtol=5e-3, atol=2e-4, err_msg=k)
    close(ag.obs_norm.mean, g["obs_mean"], rtol=1e-4, atol=1e-5)
    # std = sqrt(E[x^2] - mean^2) cancels catastrophically in fp32 for near-constant columns
    # (normalizer.py:122-127): compare variances against the magnitude that cancels
    scale = g["obs_mean"] ** 2 + g["obs_std"] ** 2
    assert np.all(np.abs(ag.obs_norm.std ** 2 - g["obs_std"] ** 2) <= 2e-6 * scale + 1e-9)
    close(ag.disc_norm.mean_abs, g["disc_mean_abs"], rtol=1e-4, atol=1e-6)
    close(ag.task.sampler.errors, g["sampler_errors"], rtol=1e-4)
    _check_summary(g, "param", {k: v.detach().numpy() for k, v in ag.model.p.items()}, rtol=2e-4, atol_scale=10)

"""GPU parity of the per-env HIP hot path (lookup, obs/disc obs, reward, done, reset, engine step)
through the C ABI, against the golden vectors of the reference and against the CPU oracle."""
import numpy as np
import pytest

from tests.util import DEFAULT_TASK, HipMotion, gload, oracle_lib, pack_pose, pack_vel, variant

pytestmark = pytest.mark.gpu
F = np.float32
FIELDS = ("root_pos", "root_rot", "root_vel", "root_ang_vel", "dof_pos", "dof_vel")
# fp32 tolerance of the transcendental chains (atan2/sin/cos/exp on device vs host libm); indices,
# clocks and done flags are compared bit-exactly
ATOL = 5e-6


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


def make_env(L, n, pose, vel, time, off, ids, hist, contact=None, hist_vel=None, dof_err_w=None):
    import torch

    st = dict(sim_pose=T(pose), sim_vel=T(vel), time=T(time, torch.float32), time_off=T(off, torch.float32), motion_id=T(ids, torch.int32),
              hist=T(hist), hist_vel=None if hist_vel is None else T(hist_vel), done=torch.zeros(n, dtype=torch.int32, device="cuda"),
              contact=None if contact is None else T(contact.astype(np.uint8)), ref_pose=torch.zeros(n, 36, device="cuda"),
              ref_vel=torch.zeros(n, 36, device="cuda"), ret_acc=torch.zeros(n, device="cuda"), len_acc=torch.zeros(n, dtype=torch.int32, device="cuda"),
              dof_err_w=None if dof_err_w is None else T(dof_err_w))
    c = L.EnvT(n, *[L.ptr(st[k]) for k in ("sim_pose", "sim_vel", "time", "time_off", "motion_id", "hist", "hist_vel", "done", "contact", "ref_pose", "ref_vel", "ret_acc", "len_acc", "dof_err_w")])
    return st, c


def make_out(L, n, task):
    import torch

    o = dict(obs=torch.full((n, task.obs_stride), 7.0, device="cuda"), obs2=torch.full((n, task.obs_stride), 7.0, device="cuda"),
             tmo=torch.full((n, task.obs_stride), 7.0, device="cuda"), disc=torch.full((n, task.disc_stride), 7.0, device="cuda"), demo=torch.full((n, task.disc_stride), 7.0, device="cuda"),
             reward=torch.zeros(n, device="cuda"), done=torch.zeros(n, dtype=torch.int32, device="cuda"),
             mid=torch.zeros(n, dtype=torch.int32, device="cuda"), mtime=torch.zeros(n, device="cuda"), ep=torch.zeros(3, device="cuda"))
    c = L.StepOutT(*[L.ptr(o[k]) for k in ("obs", "obs2", "tmo", "disc", "demo", "reward", "done", "mid", "mtime", "ep")])
    return o, c


@pytest.mark.parametrize("n_big,vname", [(16384 + 37, "default"), (65536 + 5, "default"), (270000, "default"), (32768 + 3, "vel_phase")])
def test_env_step_grouped_path_is_identical_to_one_env_per_wave(n_big, vname):
    """From 16 384 envs on, a wave owns a GROUP of 2-16 consecutive envs (rows of the next env prefetched, reward inputs
    parked per slot, one lane per env for the scalar phase; ragged last group).  Tiling the 32 golden envs to a ragged
    large count must reproduce, env by env and bit for bit, what the one-env-per-wave launch gives for the 32."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import make_task

    v = variant(gload("obs_reward_done"), vname)
    task = make_task({**DEFAULT_TASK, **VARIANTS[vname]}, 0.01)
    mot = HipMotion()
    n = v["time"].shape[0]
    pose = pack_pose(v["root_pos"], v["root_rot"], v["dof_pos"])
    vel = pack_vel(v["root_vel"], v["root_ang_vel"], v["dof_vel"])
    head = int(v["hist_head"])
    with_vel = bool(task.enable_vel_obs)
    hv = fixture_hist_vel(v) if with_vel else None
    idx = np.arange(n_big) % n

    def run(sel):
        st, env = make_env(L, len(sel), pose[sel], vel[sel], v["time"][sel], v["time_off"][sel], v["motion_ids"][sel], fixture_hist(v)[sel],
                           v["contact"][sel], None if hv is None else hv[sel])
        o, out = make_out(L, len(sel), task)
        L.call("addhip_env_step", mot.c, task, env, out, head, L.current_stream())
        torch.cuda.synchronize()
        return st, o

    st0, o0 = run(np.arange(n))
    st1, o1 = run(idx)
    sel = torch.tensor(idx, device="cuda")
    for k in ("obs", "disc", "demo", "reward", "done", "mid", "mtime", "tmo"):
        assert torch.equal(o1[k], o0[k][sel]), k
    for k in ("time", "done", "hist", "ref_pose", "ref_vel", "ret_acc", "len_acc") + (("hist_vel",) if with_vel else ()):
        assert torch.equal(st1[k], st0[k][sel]), k
    fin = (v["done"] != 0)[idx]
    assert float(o1["ep"][2]) == fin.sum()


@pytest.mark.parametrize("n_small", [1, 3, 17])
def test_env_step_tiny_and_ragged_env_counts_and_refused_calls(n_small):
    """The smallest launches (one env; counts that fill neither a wave's group nor a workgroup) give, env by env and bit for bit, what the
    32-env golden launch gives for the same envs; nothing past the last env is touched; calls without envs or tables are refused."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import make_task

    v = variant(gload("obs_reward_done"), "default")
    task = make_task(DEFAULT_TASK, 0.01)
    mot = HipMotion()
    n = v["time"].shape[0]
    pose = pack_pose(v["root_pos"], v["root_rot"], v["dof_pos"])
    vel = pack_vel(v["root_vel"], v["root_ang_vel"], v["dof_vel"])
    head = int(v["hist_head"])

    def run(sel, count):
        # buffers hold len(sel) envs, the launch covers the first `count`
        st, env = make_env(L, len(sel), pose[sel], vel[sel], v["time"][sel], v["time_off"][sel], v["motion_ids"][sel], fixture_hist(v)[sel], v["contact"][sel])
        env.num_envs = count
        o, out = make_out(L, len(sel), task)
        L.call("addhip_env_step", mot.c, task, env, out, head, L.current_stream())
        torch.cuda.synchronize()
        return st, o, env, out

    st0, o0, _, _ = run(np.arange(n), n)
    sel = np.arange(n)[-(n_small + 2):]  # the last envs of the fixture, two more than the launch covers
    st1, o1, env, out = run(sel, n_small)
    at = torch.tensor(sel[:n_small], device="cuda")
    for k in ("obs", "disc", "demo", "reward", "done", "mid", "mtime"):
        assert torch.equal(o1[k][:n_small], o0[k][at]), k
    for k in ("time", "done", "hist", "ref_pose", "ref_vel", "ret_acc", "len_acc"):
        assert torch.equal(st1[k][:n_small], st0[k][at]), k
    # the two envs past the launch: untouched (outputs keep their fill value, clocks their input)
    assert float(o1["obs"][n_small:].min()) == 7.0 and float(o1["disc"][n_small:].min()) == 7.0 and int(o1["done"][n_small:].abs().sum()) == 0
    assert np.array_equal(st1["time"][n_small:].cpu().numpy(), v["time"][sel][n_small:])
    lib = L.load()
    import ctypes as C
    env.num_envs = 0
    assert lib.addhip_env_step(C.byref(mot.c), C.byref(task), C.byref(env), C.byref(out), head, None) != 0 and lib.addhip_last_error()
    env.num_envs = n_small
    assert lib.addhip_env_step(None, C.byref(task), C.byref(env), C.byref(out), head, None) != 0
    assert lib.addhip_env_step(C.byref(mot.c), C.byref(task), C.byref(env), C.byref(out), 3, None) != 0  # ring slot out of range


def fixture_hist(v, prefix=""):
    return np.concatenate([v[prefix + "hist_root_pos"], v[prefix + "hist_root_rot"], v[prefix + "hist_dof_pos"]], axis=-1).astype(F)


def fixture_hist_vel(v, prefix=""):
    z = np.zeros(v[prefix + "hist_root_vel"].shape[:-1] + (1,), F)
    return np.concatenate([v[prefix + "hist_root_vel"], v[prefix + "hist_root_ang_vel"], v[prefix + "hist_dof_vel"], z], axis=-1).astype(F)


VARIANTS = {"default": {}, "local": dict(global_obs=False), "noheight": dict(root_height_obs=False),
            "local_noheight": dict(global_obs=False, root_height_obs=False),
            "vel_phase": dict(enable_vel_obs=True, enable_phase_obs=True, num_phase_encoding=4),
            "local_vel": dict(global_obs=False, enable_vel_obs=True),
            # task.num_disc_obs_steps = 2 (add_observation.py:276-294, 362-375; fixture obs_reward_done_s2): the ring depth is a task field
            "two_steps": dict(num_disc_obs_steps=2), "two_steps_local_vel": dict(num_disc_obs_steps=2, global_obs=False, enable_vel_obs=True),
            # = 4 (fixture obs_reward_done_s4): the third older history / clip row is staged in the kernel's second pass
            "four_steps": dict(num_disc_obs_steps=4), "four_steps_local_vel": dict(num_disc_obs_steps=4, global_obs=False, enable_vel_obs=True)}


@pytest.mark.parametrize("vname", list(VARIANTS) + ["joint_w"])
def test_env_step_matches_reference(vname):
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import make_task

    # "joint_w": non-uniform task.joint_err_w (add_reward.py:24-52), its own fixture; same states as "default"
    from tests.test_oracle_vs_golden import step_fixture

    v = variant(gload(step_fixture(vname)), vname)
    task = make_task({**DEFAULT_TASK, **VARIANTS.get(vname, {})}, 0.01)
    mot = HipMotion()
    n = v["time"].shape[0]
    pose = pack_pose(v["root_pos"], v["root_rot"], v["dof_pos"])
    vel = pack_vel(v["root_vel"], v["root_ang_vel"], v["dof_vel"])
    head = int(v["hist_head"])
    with_vel = bool(task.enable_vel_obs)
    assert fixture_hist(v).shape[1] == task.num_disc_obs_steps and task.disc_dim == task.num_disc_obs_steps * (38 + (35 if with_vel else 0))
    st, env = make_env(L, n, pose, vel, v["time"], v["time_off"], v["motion_ids"], fixture_hist(v), v["contact"],
                       fixture_hist_vel(v) if with_vel else None, v["dof_err_w"] if vname == "joint_w" else None)
    o, out = make_out(L, n, task)
    L.call("addhip_env_step", mot.c, task, env, out, head, L.current_stream())
    torch.cuda.synchronize()
    obs = o["obs"].cpu().numpy()
    dd = task.disc_dim
    assert v["obs"].shape[1] == task.obs_dim and v["disc_obs"].shape[1] == dd
    assert obs.shape[1] == task.obs_stride and np.all(obs[:, task.obs_dim:] == 0)
    np.testing.assert_allclose(obs[:, :task.obs_dim], v["obs"], rtol=0, atol=ATOL)
    assert np.array_equal(o["obs2"].cpu().numpy(), obs)
    np.testing.assert_allclose(o["disc"].cpu().numpy()[:, :dd], v["disc_obs"], rtol=0, atol=ATOL)
    np.testing.assert_allclose(o["demo"].cpu().numpy()[:, :dd], v["disc_obs_demo"], rtol=0, atol=ATOL)
    assert np.all(o["disc"].cpu().numpy()[:, dd:] == 0)
    if with_vel:
        assert np.array_equal(st["hist_vel"].cpu().numpy()[:, head], vel)
    np.testing.assert_allclose(o["reward"].cpu().numpy(), v["reward"], rtol=0, atol=ATOL)
    # bit-exact: flags, clock, reference rows (pure gathers), recorded motion times
    assert np.array_equal(o["done"].cpu().numpy(), v["done"])
    assert np.array_equal(st["done"].cpu().numpy(), v["done"])
    # pre-reset obs rows are parked for exactly the envs that ran into the episode time limit
    tmo, timed = o["tmo"].cpu().numpy(), v["done"] == 3
    assert timed.sum() > 0 and np.array_equal(tmo[timed], obs[timed]) and np.all(tmo[~timed] == 7.0)
    assert np.array_equal(st["time"].cpu().numpy(), v["time_post"])
    assert np.array_equal(st["ref_pose"].cpu().numpy(), pack_pose(v["ref_root_pos"], v["ref_root_rot"], v["ref_dof_pos"]))
    assert np.array_equal(st["ref_vel"].cpu().numpy()[:, :35], pack_vel(v["ref_root_vel"], v["ref_root_ang_vel"], v["ref_dof_vel"])[:, :35])
    assert np.array_equal(o["mtime"].cpu().numpy(), (v["time_post"] + v["time_off"]).astype(F))
    # ring: the new state landed in slot `head`, the other two slots are untouched
    h = st["hist"].cpu().numpy()
    assert np.array_equal(h[:, head], pose)
    assert h.shape[1] == task.num_disc_obs_steps
    for s in range(h.shape[1]):
        if s != head:
            assert np.array_equal(h[:, s], fixture_hist(v)[:, s])
    # return tracker: finished episodes were folded into ep_stats and cleared
    fin = v["done"] != 0
    ep = o["ep"].cpu().numpy()
    assert ep[2] == fin.sum()
    np.testing.assert_allclose(ep[0], v["reward"][fin].sum(), rtol=1e-5)
    assert np.all(st["len_acc"].cpu().numpy()[fin] == 0) and np.all(st["len_acc"].cpu().numpy()[~fin] == 1)


def _mid_bin_u(cdf, k):
    lo = 0.0 if k == 0 else float(cdf[k - 1])
    return F(0.5 * (lo + float(cdf[k])))


@pytest.mark.parametrize("tag", ["one", "two", "one_two_steps", "one_four_steps"])
def test_env_reset_matches_reference(tag):
    """(`one_two_steps` / `one_four_steps`: task.num_disc_obs_steps = 2 / 4, fixtures reset_s2 / reset_s4 -- the reset fills an S-deep ring
    with the clip frames t-(S-1)dt .. t.)"""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import make_task
    from oracle.task import SegmentSampler

    steps = {"one_two_steps": 2, "one_four_steps": 4}.get(tag, 3)
    s2 = steps != 3
    v = variant(gload(f"reset_s{steps}" if s2 else "reset"), "one" if s2 else tag)
    two = tag == "two"
    task = make_task({**DEFAULT_TASK, "num_disc_obs_steps": steps}, 0.01)
    mot = HipMotion(two)
    n = v["time"].shape[0]
    pose = pack_pose(v["sim_root_pos"], v["sim_root_rot"], v["sim_dof_pos"])
    vel = pack_vel(v["sim_root_vel"], v["sim_root_ang_vel"], v["sim_dof_vel"])
    st, env = make_env(L, n, pose, vel, v["time"], v["time_off"], v["motion_ids"], fixture_hist(v))
    env_ids = v["env_ids"]
    done = np.zeros(n, np.int32)
    done[env_ids] = 1
    st["done"].copy_(T(done))
    # the reference's multinomial / rand draws, re-expressed as the uniforms of the inverse-CDF sampler
    weights = np.asarray([1.0, 3.0] if two else [1.0], F)
    weights /= weights.sum()
    clip_cdf = np.cumsum(weights).astype(F)
    orc = SegmentSampler(mot.lengths, 0.01, 20, None, 0.02)
    orc.errors = v["sampler_errors"].copy()
    probs = orc.probs(v["draw_ids"])
    u_clip, u_seg, u_jit = np.zeros(n, F), np.zeros(n, F), np.zeros(n, F)
    for j, e in enumerate(env_ids):
        u_clip[e] = _mid_bin_u(clip_cdf, int(v["draw_ids"][j]))
        u_seg[e] = _mid_bin_u(np.cumsum(probs[j]), int(v["draw_segments"][j]))
        u_jit[e] = v["draw_jitter"][j]
    smp = dict(errors=T(v["sampler_errors"]), seg=T(orc.segment_sizes), cdf=T(clip_cdf), bits=torch.zeros(1, dtype=torch.int32, device="cuda"),
               es=torch.zeros(orc.errors.size, device="cuda"), ec=torch.zeros(orc.errors.size, device="cuda"))
    sc = L.SamplerT(L.ptr(smp["errors"]), L.ptr(smp["seg"]), L.ptr(smp["cdf"]), 20, -1.0, 0.02, 1, L.ptr(smp["bits"]), L.ptr(smp["es"]), L.ptr(smp["ec"]))
    obs = torch.zeros(n, task.obs_stride, device="cuda")
    disc = torch.zeros(n, task.disc_stride, device="cuda")
    demo = torch.zeros(n, task.disc_stride, device="cuda")
    head = int(v["hist_head"])
    L.call("addhip_env_reset", mot.c, task, env, sc, P(T(u_clip)), P(T(u_seg)), P(T(u_jit)), L.ptr(obs), L.ptr(disc), L.ptr(demo), 0, head,
           L.current_stream())
    torch.cuda.synchronize()
    assert np.array_equal(st["motion_id"].cpu().numpy(), v["post_motion_ids"])
    assert np.array_equal(st["time"].cpu().numpy(), v["post_time"])
    assert np.array_equal(st["time_off"].cpu().numpy(), v["post_time_off"])  # quantised start times: bit-exact
    assert np.all(st["done"].cpu().numpy() == 0)
    post_pose = pack_pose(v["post_sim_root_pos"], v["post_sim_root_rot"], v["post_sim_dof_pos"])
    post_vel = pack_vel(v["post_sim_root_vel"], v["post_sim_root_ang_vel"], v["post_sim_dof_vel"])
    assert np.array_equal(st["sim_pose"].cpu().numpy(), post_pose)  # set_qpos payload == table rows
    assert np.array_equal(st["sim_vel"].cpu().numpy()[:, :35], post_vel[:, :35])
    assert np.array_equal(st["hist"].cpu().numpy(), fixture_hist(v, "post_"))
    np.testing.assert_allclose(obs.cpu().numpy()[env_ids][:, :264], v["obs"][env_ids], rtol=0, atol=ATOL)
    dd = task.disc_dim
    assert dd == 38 * steps and v["disc_obs"].shape[1] == dd
    np.testing.assert_allclose(disc.cpu().numpy()[env_ids][:, :dd], v["disc_obs"][env_ids], rtol=0, atol=ATOL)
    np.testing.assert_allclose(demo.cpu().numpy()[env_ids][:, :dd], v["disc_obs_demo"][env_ids], rtol=0, atol=ATOL)
    untouched = np.setdiff1d(np.arange(n), env_ids)
    assert np.all(obs.cpu().numpy()[untouched] == 0)


def test_lookup_bit_exact():
    import torch
    import add_gym_amd._lib as L

    g = gload("lookup")
    for two, ids, times, ref in ((False, g["ids"], g["times"], g["idx"]), (True, g["two_ids"], g["two_times"], g["two_idx"])):
        mot = HipMotion(two)
        n = len(ids)
        idx = torch.zeros(n, dtype=torch.int32, device="cuda")
        pose = torch.zeros(n, 36, device="cuda")
        L.call("addhip_motion_lookup", mot.c, P(T(ids, torch.int32)), P(T(times, torch.float32)), n, L.ptr(idx), L.ptr(pose), None, L.current_stream())
        torch.cuda.synchronize()
        assert np.array_equal(idx.cpu().numpy(), ref)
        assert np.array_equal(pose.cpu().numpy(), mot.pose.cpu().numpy()[ref])
    # the corrected multi-clip mode differs from the reference's quirk and never leaves a clip's rows
    fixed = HipMotion(True, reference_compat=False)
    idx = torch.zeros(len(g["two_ids"]), dtype=torch.int32, device="cuda")
    L.call("addhip_motion_lookup", fixed.c, P(T(g["two_ids"], torch.int32)), P(T(g["two_times"], torch.float32)), len(idx), L.ptr(idx), None, None,
           L.current_stream())
    orc = oracle_lib(two=True, reference_compat=False)
    assert np.array_equal(idx.cpu().numpy(), orc.step_index(g["two_ids"], g["two_times"]))


def test_kin_engine_step_matches_oracle_sim():
    import torch
    import add_gym_amd._lib as L
    from oracle.loop import KinematicSim

    rng = np.random.RandomState(0)
    n = 333
    sim = KinematicSim(n, 29, 0.01)
    sim.dof_pos[:] = rng.standard_normal((n, 29)).astype(F)
    pose = T(pack_pose(sim.root_pos, sim.root_rot, sim.dof_pos))
    vel = T(pack_vel(sim.root_vel, sim.root_ang, sim.dof_vel))
    for _ in range(3):
        act = rng.standard_normal((n, 32)).astype(F)
        L.call("addhip_kin_engine_step", L.ptr(pose), L.ptr(vel), P(T(act)), 32, n, 0.5, 0.01, L.current_stream())
        sim.step(act[:, :29])
    torch.cuda.synchronize()
    assert np.array_equal(pose.cpu().numpy()[:, 7:], sim.dof_pos)
    assert np.array_equal(vel.cpu().numpy()[:, 6:35], sim.dof_vel)


def test_env_step_full_size_properties():
    """BASELINE config size (4096 envs): spot-check 256 envs against the oracle, permutation
    equivariance over the whole batch, flags in range, pads zero."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import make_task
    from oracle import task as OT

    n = 4096
    rng = np.random.RandomState(3)
    lib = oracle_lib(golden_tables=True)
    task = make_task(DEFAULT_TASK, 0.01)
    mot = HipMotion()
    ids = np.zeros(n, np.int64)
    t_ref = rng.rand(n).astype(F) * 6.0
    rp, rr, rv, ra, dp, dv = lib.get_step(ids, t_ref)
    q = rr + rng.standard_normal((n, 4)).astype(F) * 0.05
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    sim = (rp + rng.standard_normal((n, 3)).astype(F) * 0.05, q.astype(F), rv + rng.standard_normal((n, 3)).astype(F) * 0.2,
           ra + rng.standard_normal((n, 3)).astype(F) * 0.2, dp + rng.standard_normal((n, 29)).astype(F) * 0.1,
           dv + rng.standard_normal((n, 29)).astype(F) * 0.5)
    sim = tuple(x.astype(F) for x in sim)
    time = (np.floor(rng.rand(n) * 300) * 0.01).astype(F)
    off = (t_ref - time).astype(F)
    hist = rng.standard_normal((n, 3, 36)).astype(F)
    hist[:, :, 3:7] /= np.linalg.norm(hist[:, :, 3:7], axis=-1, keepdims=True)
    contact = rng.rand(n) < 0.01
    pose, vel = pack_pose(sim[0], sim[1], sim[4]), pack_vel(sim[2], sim[3], sim[5])

    def run(perm):
        st, env = make_env(L, n, pose[perm], vel[perm], time[perm], off[perm], ids[perm], hist[perm], contact[perm])
        o, out = make_out(L, n, task)
        L.call("addhip_env_step", mot.c, task, env, out, 1, L.current_stream())
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in o.items()}

    ident = np.arange(n)
    a = run(ident)
    perm = rng.permutation(n)
    b = run(perm)
    for k in ("obs", "disc", "demo", "reward", "done"):
        assert np.array_equal(a[k][perm], b[k]), k
    assert set(np.unique(a["done"])) <= {0, 1, 2, 3}
    assert np.all(np.isfinite(a["obs"])) and np.all(a["obs"][:, 264:] == 0)
    sub = rng.choice(n, 256, replace=False)
    ts = OT.TaskState(OT.TaskCfg(), lib, 256)
    ts.time, ts.time_off, ts.motion_ids = time[sub].copy(), off[sub].copy(), ids[sub].copy()
    ts.head = 1
    ts.hist["root_pos"], ts.hist["root_rot"], ts.hist["dof_pos"] = hist[sub][:, :, 0:3].copy(), hist[sub][:, :, 3:7].copy(), hist[sub][:, :, 7:].copy()
    obs, d_obs, d_demo, r, done = ts.step(tuple(x[sub] for x in sim), contact[sub])
    np.testing.assert_allclose(a["obs"][sub][:, :264], obs, rtol=0, atol=ATOL)
    np.testing.assert_allclose(a["disc"][sub][:, :114], d_obs, rtol=0, atol=ATOL)
    np.testing.assert_allclose(a["demo"][sub][:, :114], d_demo, rtol=0, atol=ATOL)
    np.testing.assert_allclose(a["reward"][sub], r, rtol=0, atol=ATOL)
    assert np.array_equal(a["done"][sub], done)

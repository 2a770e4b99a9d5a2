"""Recorded plans and schedules on the GPU (include/addhip.h, "recorded plans"): a replay is the recorded launches, ranges replay
independently, parameter blocks are the plan's own copies, and a schedule orders sections across streams and reports buckets where the
host must issue its exchange step."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


def _setup():
    import torch
    import add_gym_amd  # noqa: F401
    from add_gym_amd import _lib as L
    from add_gym_amd.hotpath import gemm
    from add_gym_amd.learning.model import Plan, Schedule

    return torch, L, gemm, Plan, Schedule


def test_replay_equals_direct_calls_and_ranges_replay_alone():
    torch, L, gemm, Plan, _ = _setup()
    dev = "cuda"
    torch.manual_seed(0)
    M, N, K = 384, 256, 160
    A, B, bias = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.randn(N, device=dev)
    Cd, Cp = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
    sd, sp = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def calls(Cout, sout):
        g = gemm(M, N, K, L.ptr(A), K, 1, L.ptr(B), K, 1, L.ptr(Cout), N, L.EPI_BIAS_RELU, L.ptr(bias))
        return g, [("addhip_gemm_f32", (g,)), ("addhip_col_sum", (L.ptr(Cout), M, N, N, L.ptr(sout), 0.5, 0))]

    g, direct = calls(Cd, sd)
    for name, args in direct:
        L.call(name, *args, st)
    p = Plan()
    g2, rec = calls(Cp, sp)
    for name, args in rec:
        p.add(name, *args)
    g2.M = 1  # the plan replays its own copy of the descriptor, taken at record time
    del g2
    assert float(Cp.abs().sum()) == 0.0  # recording launched nothing
    p.run(st)
    torch.cuda.synchronize()
    ref = torch.relu(A @ B.t() + bias)
    assert torch.allclose(Cd, ref, atol=2e-4, rtol=1e-4)
    assert torch.equal(Cp, Cd) and torch.allclose(sp, sd, rtol=1e-5)  # (the column sum adds by float atomics: last bits vary)
    # a range replays alone: only the column sum, over a changed C
    Cp.fill_(2.0)
    p.run(st, 1, 2)
    torch.cuda.synchronize()
    assert torch.allclose(sp, torch.full_like(sp, 0.5 * 2.0 * M)) and float(Cp.min()) == 2.0
    assert [(n, [(x.M, x.N, x.K) for x in gs]) for n, gs in p.launches()] == [("addhip_gemm_f32", [(M, N, K)]), ("addhip_col_sum", [])]


def test_schedule_orders_sections_across_streams_and_reports_buckets():
    """Three sections on two streams: the second stream's section needs the first section's result (wait_before), the last section
    on the main stream needs the second stream's (wait_after on its bucket); buckets are reported in issue order with their stream."""
    torch, L, gemm, Plan, Schedule = _setup()
    dev = "cuda"
    n = 1 << 22
    x, y, z = torch.zeros(n, device=dev), torch.zeros(n, 1, device=dev), torch.zeros(4, device=dev)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    p = Plan()
    p.add("addhip_fill_normal", L.ptr(x), n, 7, 1)                       # 0: x = N(0,1) draws            (stream 0)
    p.add("addhip_col_sum", L.ptr(x), n // 4, 4, 4, L.ptr(z), 1.0, 0)    # 1: z = column sums of x[n/4,4] (stream 1, after section 0)
    p.add("addhip_fill_zero", L.ptr(x), n)                               # 2: x = 0                       (stream 0, after section 1)
    S = L.SectionT   # {stream, first, last, wait_before, wait_after, bucket}
    secs = (S * 4)(S(0, 0, 1, -1, -1, 0), S(1, 1, 2, 0, -1, 1), S(0, 2, 3, 1, -1, -1), S(0, 3, 3, -1, 1, 2))
    sched = Schedule(p, secs, 2, buckets=["a", "b", "c"])
    seen = []
    for _ in range(3):  # replays reuse the schedule's events
        seen.clear()
        x.fill_(5.0)
        sched.run([main.cuda_stream, side.cuda_stream], lambda b, si: seen.append((b, si)))
        torch.cuda.synchronize()
        assert seen == [("a", 0), ("b", 1), ("c", 0)]
        assert float(x.abs().max()) == 0.0          # section 2 ran after section 1 had read x ...
        ref = torch.zeros(n, device=dev)
        L.call("addhip_fill_normal", L.ptr(ref), n, 7, 1, main.cuda_stream)
        torch.cuda.synchronize()
        assert torch.allclose(z, ref.view(-1, 4).sum(0), atol=1e-2)   # ... and section 1 saw section 0's draws, not the 5s
    sched.run([main.cuda_stream, side.cuda_stream])  # without a call-back
    torch.cuda.synchronize()


def test_agent_update_sections_run_through_the_library_schedule():
    """The product's update step: ADDAgent._run_update_sections hands its recorded plan and eight-section schedule to
    addhip_schedule_run; gradients equal a single-stream replay of the same plan in call order."""
    torch, L, gemm, Plan, Schedule = _setup()
    from add_gym_amd.config import load_config
    from add_gym_amd.learning.add_agent import ADDAgent

    ag = ADDAgent(load_config("train", ["engine.num_envs=64", "task.motion_file=synthetic:1x300"]))
    torch.manual_seed(1)
    W = ag._W
    for k in ("norm_obs", "norm_act", "mb_adv", "mb_tar"):
        W[k].normal_()
    W["mb_logp"].fill_(ag._model.logp_const - 3.0)
    W["mb_mask"].fill_(1.0)
    W["norm_diff"][:ag.Mb].normal_()
    m = ag._model
    ag._run_update_sections()
    torch.cuda.synchronize()
    g_sched = m.grads.clone()
    assert float(g_sched.abs().sum()) > 0
    m.grads.zero_()
    W["stats"].zero_()
    ag._update_plan.run(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    # same launches, same data; only the float-atomic bias-gradient sums may differ in their last bits
    assert torch.allclose(m.grads, g_sched, rtol=1e-4, atol=1e-6 * float(g_sched.abs().max()))


def test_mlp_forward_backward_launch_directly_and_match_torch():
    """addhip_mlp_forward / addhip_mlp_backward called with a stream (not recorded): a 2-hidden-layer net on 200 rows against torch
    autograd -- activations, every weight and bias gradient; the marks count the launches."""
    torch, L, gemm, Plan, _ = _setup()
    dev = "cuda"
    torch.manual_seed(3)
    rows, in_dim, in_ld, hid = 200, 40, 48, [96, 64]
    x = torch.zeros(rows, in_ld, device=dev)
    x[:, :in_dim].normal_()
    Ws = [torch.randn(hid[0], in_ld, device=dev) * 0.2, torch.randn(hid[1], hid[0], device=dev) * 0.2]
    Ws[0][:, in_dim:] = 0
    bs = [torch.randn(h, device=dev) * 0.1 for h in hid]
    gW, gb = [torch.zeros_like(w) for w in Ws], [torch.zeros_like(b) for b in bs]
    h = [torch.zeros(rows, k, device=dev) for k in hid]
    dz = [torch.zeros(rows, k, device=dev) for k in hid]
    hb = [torch.zeros(rows, (k + 31) // 32, dtype=torch.int32, device=dev) for k in hid]
    slabs = torch.zeros(2 * 32 * hid[0] * hid[0], device=dev)
    c = L.MlpT()
    c.num_hidden, c.in_dim, c.in_ld, c.head_rows, c.precision, c.rows_cap = 2, in_dim, in_ld, 1, L.PREC_F32, rows
    for i in range(2):
        c.hidden[i] = hid[i]
        c.W[i], c.b[i], c.gW[i], c.gb[i] = L.ptr(Ws[i]), L.ptr(bs[i]), L.ptr(gW[i]), L.ptr(gb[i])
        c.h[i], c.dz[i], c.hbits[i] = L.ptr(h[i]), L.ptr(dz[i]), L.ptr(hb[i])
    c.slabs, c.slab_floats = L.ptr(slabs), slabs.numel()
    st = torch.cuda.current_stream().cuda_stream
    L.call("addhip_mlp_forward", c, L.ptr(x), None, rows, None, None, 1, None, st)
    xt = x.clone().requires_grad_(False)
    Wt = [w.clone().requires_grad_(True) for w in Ws]
    bt = [b.clone().requires_grad_(True) for b in bs]
    h0 = torch.relu(xt @ Wt[0].t() + bt[0])
    h1 = torch.relu(h0 @ Wt[1].t() + bt[1])
    torch.cuda.synchronize()
    assert torch.allclose(h[0], h0, atol=1e-4) and torch.allclose(h[1], h1, atol=1e-4)
    top = torch.randn(rows, hid[1], device=dev)
    (h1 * top).sum().backward()
    dz[1].copy_(top * (h1 > 0))  # the caller's part: d loss / d pre-activation of the last hidden layer
    marks = L.MlpMarksT()
    L.call("addhip_mlp_backward", c, L.ptr(x), None, rows, None, L.BWD_SIGN_BITS, C.byref(marks), None, st)
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.allclose(gW[i], Wt[i].grad, atol=2e-3, rtol=1e-3), i
        assert torch.allclose(gb[i], bt[i].grad, atol=2e-3, rtol=1e-3), i
    # layer 1: dW GEMM + combine, top bias column sum, zero of gb[0], dX GEMM; layer 0: dW GEMM + combine.  `early` (every gradient but
    # W[0] / b[0] is final) lies behind the column sum that writes gb[1]
    assert marks.launches == 7 and (marks.dw_first[1], marks.dw_last[1], marks.early) == (0, 2, 3) and (marks.dw_first[0], marks.dw_last[0]) == (5, 7)
    # rows beyond the workspace are refused
    assert L.load().addhip_mlp_forward(C.byref(c), L.ptr(x), None, rows + 1, None, None, 0, None, st) != 0


def test_plan_destroy_is_refused_while_a_schedule_refers_to_the_plan():
    """A schedule keeps a pointer to its plan: destroying the plan first would leave it dangling, so the library refuses (destroy order:
    schedules, then the plan) -- and a failing section still joins the side streams back into streams[0]."""
    import torch
    import add_gym_amd._lib as L

    lib = L.load()
    h, sc = C.c_void_p(), C.c_void_p()
    assert lib.addhip_plan_create(C.byref(h)) == 0
    buf = torch.ones(64, device="cuda")
    lib.addhip_plan_record_begin(h)
    assert lib.addhip_fill_zero(L.ptr(buf), 64, None) == 0
    lib.addhip_plan_record_end(h)
    secs = (L.SectionT * 1)(L.SectionT(1, 0, 1, -1, -1, -1))
    assert lib.addhip_schedule_create(h, secs, 1, 2, C.byref(sc)) == 0
    assert lib.addhip_plan_destroy(h) != 0 and b"schedule" in lib.addhip_last_error()
    side = torch.cuda.Stream()
    streams = (C.c_void_p * 2)(torch.cuda.current_stream().cuda_stream, side.cuda_stream)
    assert lib.addhip_schedule_run(sc, streams, C.cast(None, L.BUCKET_FN), None) == 0
    torch.cuda.synchronize()
    assert float(buf.abs().max()) == 0.0
    assert lib.addhip_schedule_destroy(sc) == 0 and lib.addhip_plan_destroy(h) == 0

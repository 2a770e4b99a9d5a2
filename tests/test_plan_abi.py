"""CPU-only: the recorded-plan half of the C ABI (include/addhip.h, "recorded plans").  Recording launches nothing -- an entry point
called between record_begin / record_end checks its arguments and appends itself to the plan -- so the bookkeeping, the copies of the
parameter blocks and the error paths are testable without a GPU.  Replays are covered by tests/test_hip_plan.py (-m gpu)."""
import ctypes as C

import pytest


def _lib():
    import add_gym_amd  # noqa: F401
    from add_gym_amd import _lib as L

    return L, L.load()


def _gemm(L, M=256, N=128, K=64, **kw):
    g = L.GemmT()
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.a_kcontig = 0x10000, K, 1
    g.B, g.ldb, g.b_kcontig = 0x20000, K, 1
    g.C, g.ldc = 0x30000, N
    g.split_k, g.alpha = 1, 1.0
    for k, v in kw.items():
        setattr(g, k, v)
    return g


def test_recording_appends_calls_and_copies_descriptors():
    L, lib = _lib()
    h = C.c_void_p()
    assert lib.addhip_plan_create(C.byref(h)) == 0
    assert lib.addhip_plan_size(h) == 0
    assert lib.addhip_plan_record_begin(h) == 0
    assert lib.addhip_plan_record_begin(h) != 0 and b"already recording" in lib.addhip_last_error()
    g = _gemm(L)
    assert lib.addhip_fill_zero(0x40000, 1024, None) == 0          # recorded, not launched (there is no GPU here)
    assert lib.addhip_gemm_f32(C.byref(g), None) == 0
    g.M = 7                                                          # the plan holds its own copy
    pair = (L.GemmT * 2)(_gemm(L, N=256), _gemm(L, N=256, C=0x50000))
    assert lib.addhip_gemm_grouped(pair, 2, None) == 0
    # argument checks still run while recording: a bad call is refused and not appended
    assert lib.addhip_fill_zero(None, 1024, None) != 0
    bad = _gemm(L, M=0)
    assert lib.addhip_gemm_f32(C.byref(bad), None) != 0
    assert lib.addhip_plan_record_end(h) == 0
    assert lib.addhip_plan_record_end(h) != 0
    assert lib.addhip_plan_size(h) == 3
    assert [lib.addhip_plan_call_name(h, i) for i in range(4)] == [b"addhip_fill_zero", b"addhip_gemm_f32", b"addhip_gemm_grouped", None]
    out = (L.GemmT * 4)()
    assert lib.addhip_plan_call_gemms(h, 0, out, 4) == 0
    assert lib.addhip_plan_call_gemms(h, 1, out, 4) == 1 and (out[0].M, out[0].N, out[0].K, out[0].C) == (256, 128, 64, 0x30000)
    assert lib.addhip_plan_call_gemms(h, 2, out, 4) == 2 and (out[1].N, out[1].C) == (256, 0x50000)
    assert lib.addhip_plan_call_gemms(h, 2, out, 1) < 0 and lib.addhip_plan_call_gemms(h, 9, out, 4) < 0
    # ranges are validated before anything runs; an empty range is a no-op
    assert lib.addhip_plan_run(h, 0, 0, None) == 0 and lib.addhip_plan_run(h, 3, 3, None) == 0
    assert lib.addhip_plan_run(h, 2, 1, None) != 0 and lib.addhip_plan_run(h, 0, 4, None) != 0
    assert lib.addhip_plan_destroy(h) == 0


def test_schedule_refuses_sections_that_cannot_be_issued_in_order():
    L, lib = _lib()
    h = C.c_void_p()
    assert lib.addhip_plan_create(C.byref(h)) == 0
    lib.addhip_plan_record_begin(h)
    for _ in range(4):
        assert lib.addhip_fill_zero(0x40000, 16, None) == 0
    lib.addhip_plan_record_end(h)
    sc = C.c_void_p()
    S = L.SectionT
    cases = {
        b"names stream": [S(2, 0, 1, -1, -1, -1)],                      # two streams: index 2 does not exist
        b"covers": [S(0, 0, 5, -1, -1, -1)],                             # past the plan's four calls
        b"waits for a later section": [S(0, 0, 1, 1, -1, -1), S(1, 1, 2, -1, -1, -1)],
    }
    for text, secs in cases.items():
        arr = (S * len(secs))(*secs)
        assert lib.addhip_schedule_create(h, arr, len(secs), 2, C.byref(sc)) != 0
        assert text in lib.addhip_last_error(), (text, lib.addhip_last_error())
    assert lib.addhip_plan_destroy(h) == 0


def test_python_plan_records_into_a_library_plan():
    """learning.model.Plan: the builder the agent uses; add() records at once, a refused call leaves the plan and the thread usable."""
    L, lib = _lib()
    from add_gym_amd.learning.model import Plan

    p = Plan()
    assert p.add("addhip_fill_zero", 0x40000, 64) == 0
    g = _gemm(L)
    assert p.add("addhip_gemm_f32", g) == 1 and len(p) == 2
    with pytest.raises(L.AddhipError):
        p.add("addhip_fill_zero", None, 64)
    assert p.add("addhip_fill_zero", 0x40000, 64) == 2 and len(p) == 3
    names = [(n, [(x.M, x.N, x.K) for x in gs]) for n, gs in p.launches()]
    assert names == [("addhip_fill_zero", []), ("addhip_gemm_f32", [(256, 128, 64)]), ("addhip_fill_zero", [])]


def test_composite_entry_points_record_their_launches():
    """addhip_mlp_forward called while recording appends one GEMM per layer (two where a few rows past a multiple of 128 are split off),
    with the descriptors csrc/learner.hip documents; addhip_update_schedule lays ten sections over the marks of the two loss sections."""
    L, lib = _lib()
    from add_gym_amd.learning.model import Plan

    c = L.MlpT()
    c.num_hidden, c.in_dim, c.in_ld, c.head_rows, c.precision, c.rows_cap = 2, 114, 128, 1, L.PREC_F32, 1025
    for i, h in enumerate((256, 128)):
        c.hidden[i] = h
        c.W[i], c.b[i], c.h[i], c.dz[i], c.hbits[i] = 0x100000 * (i + 1), 0x900000 + 0x1000 * i, 0x2000000 * (i + 1), 0x6000000 * (i + 1), 0xA000000 + 0x100000 * i
    p = Plan()
    p.add("addhip_mlp_forward", c, 0x50000000, None, 1025, None, None, 1, None)
    shapes = [(n, [(x.M, x.N, x.K, x.epilogue, bool(x.relu_bits)) for x in gs]) for n, gs in p.launches()]
    assert shapes == [("addhip_gemm_f32", [(1024, 256, 128, L.EPI_BIAS_RELU, True)]), ("addhip_gemm_f32", [(1, 256, 128, L.EPI_BIAS_RELU, False)]),
                      ("addhip_gemm_f32", [(1024, 128, 256, L.EPI_BIAS_RELU, True)]), ("addhip_gemm_f32", [(1, 128, 256, L.EPI_BIAS_RELU, False)])]
    with pytest.raises(L.AddhipError):  # more rows than the workspace holds
        p.add("addhip_mlp_forward", c, 0x50000000, None, 1026, None, None, 0, None)
    pm, dm = L.PpoMarksT(60, 30, 20, 50), L.DiscMarksT(40, 10, 12, 20, 21, 24)
    secs = (L.SectionT * 10)()
    assert lib.addhip_update_schedule(5, C.byref(pm), C.byref(dm), secs, 10) == 10
    got = [(s.stream, s.first, s.last, s.wait_before, s.wait_after, s.bucket) for s in secs]
    assert got == [(0, 5, 25, -1, -1, 0), (1, 35, 55, -1, -1, 1), (1, 55, 65, -1, -1, -1), (0, 25, 35, -1, 2, 3), (2, 65, 75, -1, -1, -1),
                   (3, 75, 77, 4, -1, -1), (2, 77, 85, -1, -1, -1), (3, 85, 86, -1, -1, -1), (3, 86, 89, 6, -1, -1), (2, 89, 105, 7, 8, 2)]
    assert lib.addhip_update_schedule(5, C.byref(pm), C.byref(dm), secs, 9) != 0


def _two_layer_net(L, slab_floats):
    c = L.MlpT()
    c.num_hidden, c.in_dim, c.in_ld, c.head_rows, c.precision, c.rows_cap = 2, 500, 512, 1, L.PREC_F32, 1024
    for i, h in enumerate((256, 128)):
        c.hidden[i] = h
        c.W[i], c.b[i], c.h[i], c.dz[i], c.hbits[i] = 0x100000 * (i + 1), 0x900000 + 0x1000 * i, 0x2000000 * (i + 1), 0x6000000 * (i + 1), 0xA000000 + 0x100000 * i
        c.gW[i], c.gb[i] = 0x1100000 * (i + 1), 0x1900000 + 0x1000 * i
    c.slabs, c.slab_floats = 0x70000000, slab_floats
    return c


def test_a_composite_call_refused_between_its_launches_leaves_the_plan_unchanged():
    """addhip_mlp_backward checks each layer's split-K scratch when it reaches that layer: refused there, the launches it had already
    recorded for the layers above are dropped again (csrc/learner.hip: PlanGuard)."""
    L, lib = _lib()
    from add_gym_amd.learning.model import Plan, split_k_for

    top = split_k_for(128, 256, 1024) * 128 * 256       # the top layer's need: the first check passes ...
    low = split_k_for(256, 512, 1024) * 256 * 512
    assert low > top                                      # ... the lower layer's does not
    p = Plan()
    p.add("addhip_fill_zero", 0x40000, 64)
    with pytest.raises(L.AddhipError, match="split-K scratch"):
        p.add("addhip_mlp_backward", _two_layer_net(L, top), 0x50000000, None, 1024, None, L.BWD_TOP_BIAS_DONE, None, None)
    assert len(p) == 1
    n = p.add("addhip_mlp_backward", _two_layer_net(L, low), 0x50000000, None, 1024, None, L.BWD_TOP_BIAS_DONE, None, None)
    assert n == 1 and len(p) > 1


def test_early_mark_of_a_two_layer_net_comes_after_its_top_bias_sum():
    """With two hidden layers and the top bias gradient left to addhip_mlp_backward (flags without TOP_BIAS_DONE), `early` -- "every gradient
    but W[0] / b[0] is final" -- must lie behind the column sum that writes gb[1]."""
    L, lib = _lib()
    from add_gym_amd.learning.model import Plan, split_k_for

    c = _two_layer_net(L, 2 * split_k_for(256, 512, 1024) * 256 * 512)
    marks = L.MlpMarksT()
    p = Plan()
    p.add("addhip_mlp_backward", c, 0x50000000, None, 1024, None, 0, C.byref(marks), None)
    names = [n for n, _ in p.launches()]
    assert "addhip_col_sum" in names
    assert marks.early == names.index("addhip_col_sum") + 1 and marks.early > marks.dw_last[1]
    assert marks.launches == len(names)

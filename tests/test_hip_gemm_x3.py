"""GEMM on PLANE-STORED operands (csrc/gemm_x3.hip; include/addhip.h "plane storage"): operands and results are fp32 values kept as
three bf16 planes whose sum is the value exactly; six bf16 MFMAs per k-step give the fp32 MFMA's error bound.  Checked against float64
on the same fp32 values at the fp32 tolerance of tests/test_hip_gemm.py, every operand layout, epilogue, ragged edge, K tail and
split-K; the plane-storage outputs of the producers (GEMM epilogue, conversions, shadow refresh) bit for bit against the split restated
in tests/util.py."""
import ctypes as C

import numpy as np
import pytest

from util import from_planes, split3, to_planes

pytestmark = pytest.mark.gpu
F = np.float32
X3 = 3  # ADDHIP_STORE_BF16X3


def T(x):
    import torch

    return torch.tensor(np.ascontiguousarray(x), device="cuda")


def planes_dev(x, ld=None):
    """fp32 [rows, cols] -> device int16 tensor holding the plane storage (bit pattern of the uint16s)."""
    import torch

    return torch.tensor(to_planes(x, ld).view(np.int16), device="cuda")


def planes_host(t, cols=None):
    return from_planes(t.cpu().numpy().view(np.uint16), cols)


def test_split_restatement_is_exact():
    """(no GPU work: the numpy restatement the other tests lean on)  hi + mid + lo == x and every part is a bf16 value."""
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.standard_normal(100000).astype(F) * 10.0 ** rng.uniform(-6, 6, 100000).astype(F), np.asarray([0.0, -0.0, 1.0, -1.5, 2.0 ** -100, 1e38], F)])
    hi, mid, lo = split3(x)
    assert np.array_equal(((hi.astype(np.float64) + mid) + lo).astype(F), x)
    for p in (hi, mid, lo):
        assert not np.any(p.view(np.uint32) & 0xFFFF)
    val, _ = from_planes(to_planes(x.reshape(-1, 2)[:50000].reshape(-1, 8)))
    assert np.array_equal(val, x.reshape(-1, 2)[:50000].reshape(-1, 8))


def run_gemm_x3(M, N, K, a_kc, b_kc, epilogue=0, split_k=1, seed=0, planes_out=True, scale_a=1.0, hint=0):
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    rng = np.random.RandomState(seed)
    A = (rng.uniform(-1, 1, (M, K)) * scale_a).astype(F)
    B = rng.uniform(-1, 1, (N, K)).astype(F)
    bias = rng.uniform(-1, 1, N).astype(F)
    mask = rng.uniform(-1, 1, (M, N)).astype(F)
    lda, ldb = (K if a_kc else M), (K if b_kc else N)
    dA = planes_dev(A if a_kc else A.T.copy())
    dB = planes_dev(B if b_kc else B.T.copy())
    ldc = (N + 7) // 8 * 8
    dC = torch.full((max(split_k, 1), M, ldc), 9.0, device="cuda")
    dC16 = torch.full((M, 3 * ldc), 77, device="cuda", dtype=torch.int16)
    dbias, dmask = T(bias), T(mask)
    dcs = torch.full((N,), 0.5, device="cuda")
    use16 = planes_out and split_k <= 1 and N % 8 == 0
    g = gemm(M, N, K, L.ptr(dA), lda, a_kc, L.ptr(dB), ldb, b_kc, L.ptr(dC), ldc, epilogue, L.ptr(dbias), L.ptr(dmask), N, None, None, split_k, 1.0,
             L.ptr(dcs) if epilogue == 3 else None, L.PREC_BF16X3, operands_bf16=X3, C16=L.ptr(dC16) if use16 else None, ldc16=ldc, c16_planes=X3, hint=hint)
    L.call("addhip_gemm_f32", g, L.current_stream())
    torch.cuda.synchronize()
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    ref = A64 @ B64.T
    scale = np.abs(A64) @ np.abs(B64).T
    if epilogue in (1, 2):
        ref = ref + bias
    if epilogue == 2:
        ref = np.maximum(ref, 0)
    if epilogue == 3:
        ref = np.where(mask > 0, ref, 0)
    out = dC.cpu().numpy().astype(np.float64)
    got = out.sum(0)[:, :N] if split_k > 1 else out[0][:, :N]
    err = np.abs(got - ref)
    worst = float((err / np.maximum(scale, 1e-30)).max())
    assert np.all(err <= 4e-7 * scale + 1e-6 * scale_a), (M, N, K, a_kc, b_kc, epilogue, split_k, worst)  # the fp32 bound of tests/test_hip_gemm.py
    if ldc > N:
        assert np.all(out[0][:, N:] == 9.0)
    if use16:  # the plane copy of the result is the exact split of the fp32 result
        assert np.array_equal(dC16.cpu().numpy().view(np.uint16)[:, :3 * (N // 8) * 8], to_planes(dC[0].cpu().numpy()[:, :N]))
        if ldc > N:
            assert bool((dC16[:, 3 * N:] == 77).all())
    if epilogue == 3:
        cs = dcs.cpu().numpy().astype(np.float64) - 0.5
        cs_scale = np.where(mask > 0, scale, 0).sum(0)
        assert np.all(np.abs(cs - got.sum(0)) <= 4e-7 * cs_scale + 1e-5)
    return worst


# tile configurations of csrc/gemm_x3.hip: chosen by the dispatcher from the shape (0) or forced (ADDHIP_GEMM_HINT_*)
CONFIGS = {"auto": 0, "256x256": 1, "256x128": 32, "128x128": 2}


@pytest.mark.parametrize("config", list(CONFIGS))
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0), (0, 1)])
def test_gemm_x3_layouts_edges_epilogues_split_k(a_kc, b_kc, config):
    h = CONFIGS[config]
    M0 = 4104 if not a_kc else 4100  # m-contiguous operands move groups of 8 rows
    N0 = 1032 if not b_kc else 1000
    run_gemm_x3(M0, N0, 1024 + 8, a_kc, b_kc, hint=h)   # ragged M / N tiles, a K tail of one 8-group
    run_gemm_x3(256, 128, 64, a_kc, b_kc, hint=h)
    run_gemm_x3(128, 128, 8, a_kc, b_kc, hint=h)        # one partial stage: shorter than the ring
    run_gemm_x3(264, 136, 40, a_kc, b_kc, hint=h)       # 2.5 stages
    if config in ("auto", "256x256"):
        run_gemm_x3(16384 if not a_kc else 16385, 512, 272, a_kc, b_kc, hint=h)  # the first layers' K = 272 = 17 stages
    if a_kc:
        for epi in (1, 2, 3):
            run_gemm_x3(1000, 512, 1024, 1, b_kc, epilogue=epi, hint=h)
    if not a_kc and not b_kc:  # the weight-gradient shapes: K = minibatch rows (odd count: the discriminator's Mb + 1), split-K slabs
        run_gemm_x3(1024, 272, 16385, 0, 0, split_k=22, hint=h)
        run_gemm_x3(1024, 1024, 16384, 0, 0, split_k=8, hint=h)
        run_gemm_x3(32, 512, 4096, 0, 0, split_k=32, hint=h)


def test_gemm_x3_error_is_at_fp32_level():
    """Worst |err| / sum|a||b| of the plane-storage GEMM next to the fp32 MFMA's on the same operands (both ~3.3e-7 at K = 1024),
    also on operands of very different magnitudes (the split is exact at any exponent)."""
    from test_hip_gemm import run_gemm

    e_x3 = run_gemm_x3(2048, 1024, 1024, 1, 1, seed=5)
    e_f32 = run_gemm(2048, 1024, 1024, 1, 1, seed=5, precision=0)
    assert e_x3 <= 2.0 * e_f32 + 1e-8, (e_x3, e_f32)
    run_gemm_x3(512, 256, 512, 1, 1, seed=6, scale_a=1e-6)
    run_gemm_x3(512, 256, 512, 1, 0, seed=7, scale_a=1e5)


def test_fp32_operand_gemm_refuses_a_plane_result():
    """Plane-storage results are written by the GEMMs on plane-stored operands only (the split needs registers the other kernels' epilogues
    do not have): an fp32-operand descriptor asking for one is refused, and the caller splits the fp32 result (addhip_to_bf16x3)."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    A, B = torch.randn(256, 32, device="cuda"), torch.randn(32, 512, device="cuda")
    C_ = torch.zeros(256, 512, device="cuda")
    C16 = torch.zeros(256, 3 * 512, dtype=torch.int16, device="cuda")
    with pytest.raises(RuntimeError, match="plane-stored operands only"):
        L.call("addhip_gemm_f32", gemm(256, 512, 32, L.ptr(A), 32, 1, L.ptr(B), 512, 0, L.ptr(C_), 512, C16=L.ptr(C16), ldc16=512, c16_planes=X3), L.current_stream())


def test_to_bf16x3_and_shadow_refresh_planes():
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(1)
    x = (rng.standard_normal((777, 40)) * 10.0 ** rng.uniform(-5, 5, (777, 40))).astype(F)
    src = torch.zeros(777, 44, device="cuda")
    src[:, :40] = T(x)
    dst = torch.full((777, 3 * 48), 5, dtype=torch.int16, device="cuda")
    L.call("addhip_to_bf16x3", L.ptr(src), L.ptr(dst), 777, 40, 44, 48, L.current_stream())
    torch.cuda.synchronize()
    got = dst.cpu().numpy().view(np.uint16)
    assert np.array_equal(got[:, :120], to_planes(x)) and np.all(got[:, 120:] == 5)
    # the parameter shadow: the flat buffer split along its flat index, the listed matrices transposed row by row
    mats = [(0, 1024, 272), (1024 * 272 + 1024, 512, 1024), (1024 * 272 + 1024 + 512 * 1024 + 512, 40, 104)]
    count = mats[-1][0] + 40 * 104 + 8
    torch.manual_seed(3)
    params = torch.randn(count, device="cuda") * 0.03
    flat16 = torch.full((3 * count,), 9, device="cuda", dtype=torch.int16)
    trans16 = torch.full((3 * count,), 9, device="cuda", dtype=torch.int16)
    n = len(mats)
    L.call("addhip_shadow_refresh", L.ptr(params), L.ptr(flat16), L.ptr(trans16), count, (C.c_int64 * n)(*(m[0] for m in mats)),
           (C.c_int32 * n)(*(m[1] for m in mats)), (C.c_int32 * n)(*(m[2] for m in mats)), n, X3, L.current_stream())
    torch.cuda.synchronize()
    p = params.cpu().numpy()
    assert np.array_equal(flat16.cpu().numpy().view(np.uint16), to_planes(p.reshape(1, -1)).reshape(-1))
    t16 = trans16.cpu().numpy().view(np.uint16)
    covered = np.zeros(3 * count, bool)
    for off, r, c in mats:
        want = to_planes(p[off:off + r * c].reshape(r, c).T.copy())
        assert np.array_equal(t16[3 * off:3 * (off + r * c)].reshape(c, 3 * r), want)
        covered[3 * off:3 * (off + r * c)] = True
    assert np.all(t16[~covered] == 9)
    with pytest.raises(RuntimeError):  # plane storage needs whole 8-value groups
        L.call("addhip_shadow_refresh", L.ptr(params), L.ptr(flat16), L.ptr(trans16), count, (C.c_int64 * 1)(0), (C.c_int32 * 1)(12), (C.c_int32 * 1)(8), 1, X3,
               L.current_stream())

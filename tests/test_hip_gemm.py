"""fp32 MFMA GEMM (addhip_gemm_f32) against a float64 reference, all operand layouts,
epilogues, fused input normalisation, split-K slabs, ragged shapes."""
import numpy as np
import pytest

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


def T(x):
    import torch

    return torch.tensor(np.ascontiguousarray(x), device="cuda")


# per-product error bound relative to sum |a||b|: fp32 MFMA and the exact bf16x3 split share the fp32 bound; plain bf16
# truncates both operands to 8 significant bits
# (4 = ADDHIP_PREC_F16X2: 22-bit operands on per-tensor scales, all four products: representation error <= 2^-22 per operand, random in
# sign, next to the same fp32 accumulation -- held to the fp32 tolerance)
TOL = {0: 4e-7, 3: 4e-7, 4: 4e-7, 2: 2.0 ** -14, 1: 2.0 ** -6}


def amax_slots(t):
    """ADDHIP_AMAX_SLOTS float bit patterns bounding max |t| (addhip_amax_f32): what an ADDHIP_PREC_F16X2 GEMM scales operand t by."""
    import torch
    import add_gym_amd._lib as L

    slots = torch.zeros(L.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    L.call("addhip_amax_f32", L.ptr(t), t.numel(), L.ptr(slots), L.current_stream())
    _KEEP.append(slots)
    return slots


def run_gemm(M, N, K, a_kc, b_kc, epilogue=0, split_k=1, norm=False, alpha=1.0, seed=0, precision=0, accumulate=0, scale_a=1.0, scale_b=1.0, check_amax=False, hint=0):
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    rng = np.random.RandomState(seed)
    A = (rng.uniform(-1, 1, (M, K)) * scale_a).astype(F)
    B = (rng.uniform(-1, 1, (N, K)) * scale_b).astype(F)
    bias = (rng.uniform(-1, 1, N) * scale_a * scale_b).astype(F)
    mask = rng.uniform(-1, 1, (M, N)).astype(F)
    mean = rng.uniform(-1, 1, K).astype(F)
    std = rng.uniform(0.5, 2, K).astype(F)
    lda = K if a_kc else M
    ldb = K if b_kc else N
    dA = T(A if a_kc else A.T.copy())
    dB = T(B if b_kc else B.T.copy())
    ldc = (N + 3) // 4 * 4
    dC = torch.full((1 if accumulate else max(split_k, 1), M, ldc), 9.0, device="cuda")
    dbias, dmask, dmean, dstd = T(bias), T(mask), T(mean), T(std)
    dcs = torch.full((N,), 0.5, device="cuda")  # MASK epilogue also accumulates the column sums (bias gradient) here
    g = gemm(M, N, K, L.ptr(dA), lda, a_kc, L.ptr(dB), ldb, b_kc, L.ptr(dC), ldc, epilogue, L.ptr(dbias), L.ptr(dmask), N,
             L.ptr(dmean) if norm else None, L.ptr(dstd) if norm else None, split_k, alpha, L.ptr(dcs) if epilogue == 3 else None, precision,
             accumulate=accumulate, hint=hint)
    if precision == 4 and not norm:  # the fp16 split scales each operand by its tracked maximum
        g.a_amax, g.b_amax = L.ptr(amax_slots(dA)), L.ptr(amax_slots(dB))
    out_amax = torch.zeros(L.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    if check_amax:
        g.amax_out = L.ptr(out_amax)
    L.call("addhip_gemm_f32", g, L.current_stream())
    torch.cuda.synchronize()
    A64 = A.astype(np.float64)
    if norm:
        A64 = ((A - mean) / std).astype(np.float64)
    ref = alpha * (A64 @ B.astype(np.float64).T)
    scale = np.abs(A64) @ np.abs(B.astype(np.float64)).T
    if epilogue in (1, 2):
        ref = ref + bias
    if epilogue == 2:
        ref = np.maximum(ref, 0)
    if epilogue == 3:
        ref = np.where(mask > 0, ref, 0)
    out = dC.cpu().numpy().astype(np.float64)
    got = out.sum(0)[:, :N] if split_k > 1 and not accumulate else out[0][:, :N]
    if accumulate:  # every K slice added its partial product into the prefilled C (hardware atomics)
        got = got - 9.0
    err = np.abs(got - ref)
    tol = TOL[precision]
    if epilogue == 3 and precision == 1:  # a sign flip of a near-zero masked value is not an error of the product
        pass
    assert np.all(err <= tol * scale * max(1.0, abs(alpha)) + 1e-6 * scale_a * scale_b), (M, N, K, a_kc, b_kc, epilogue, split_k, precision, float((err / scale).max()))
    if check_amax:  # the tracked maximum of the result is exactly max |C| over the entries written
        assert float(out_amax.cpu().numpy().view(np.float32).max()) == float(np.abs(out[0][:, :N]).max())
    if ldc > N:
        assert np.all(out[0][:, N:] == 9.0)  # pad columns are never written
    if epilogue == 3:
        cs = dcs.cpu().numpy().astype(np.float64) - 0.5
        cs_scale = np.where(mask > 0, scale, 0).sum(0)
        assert np.all(np.abs(cs - got.sum(0)) <= 4e-7 * cs_scale + 1e-5), float(np.abs(cs - got.sum(0)).max())
    return float((err / np.maximum(scale, 1e-30)).max())


@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("shape", [(256, 128, 64), (300, 200, 264), (128, 1024, 116), (64, 32, 512), (4100, 512, 1024)])
def test_gemm_layouts(shape, a_kc, b_kc):
    M, N, K = shape
    if not a_kc:
        M = (M + 3) // 4 * 4
    run_gemm(M, N, K, a_kc, b_kc)


@pytest.mark.parametrize("epi", [1, 2, 3])
def test_gemm_epilogues(epi):
    run_gemm(515, 1024, 264, 1, 1, epilogue=epi)
    run_gemm(129, 32, 512, 1, 1, epilogue=epi)  # 29/32-wide head shape
    run_gemm(257, 512, 1024, 1, 0, epilogue=epi)


@pytest.mark.parametrize("precision", [4, 3, 2])
@pytest.mark.parametrize("config", ["256x256", "256x128"])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_split_256_row_tiles(a_kc, b_kc, config, precision):
    """The 8-wave forms of the split kernel (gemm_split_big_kernel: 256x256 and 256x128 tiles), forced on shapes the dispatcher would also give
    to the 128x128 kernel: every operand layout, ragged edges, K tails and K shorter than the register ring, epilogues, split-K."""
    h = {"256x256": 1, "256x128": 32}[config]
    run_gemm(520, 392, 1024 + 12, a_kc, b_kc, precision=precision, hint=h)
    run_gemm(256, 256, 16, a_kc, b_kc, precision=precision, hint=h)
    run_gemm(4100, 1000, 272, a_kc, b_kc, precision=precision, hint=h, check_amax=True)
    if a_kc:
        for epi in (1, 2, 3):
            run_gemm(1000, 512, 520, 1, b_kc, epilogue=epi, precision=precision, hint=h, check_amax=True)
    if not a_kc and not b_kc:
        run_gemm(1024, 272, 4096 + 4, 0, 0, split_k=5, precision=precision, hint=h)


@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_f16x2_split(a_kc, b_kc):
    """ADDHIP_PREC_F16X2 (two-way fp16 split on per-tensor power-of-two scales, four products) at the fp32 tolerance: every layout, ragged
    edges and K tails, the epilogues, split-K, operands whose magnitudes are far outside fp16's range, the tracked maximum of the result."""
    run_gemm(4100, 1000, 1024 + 12, a_kc, b_kc, precision=4)
    run_gemm(16384, 512, 272, a_kc, b_kc, precision=4, check_amax=True)
    run_gemm(2048, 1024, 512, a_kc, b_kc, precision=4, scale_a=3e-9, scale_b=7e4, check_amax=True)   # gradients x large weights
    run_gemm(2048, 1024, 512, a_kc, b_kc, precision=4, scale_a=2e7, scale_b=1e-12)
    if a_kc:
        for epi in (1, 2, 3):
            run_gemm(4096, 512, 1024, 1, b_kc, epilogue=epi, precision=4, check_amax=True)
    if not a_kc and not b_kc:
        run_gemm(1024, 1024, 16384, 0, 0, split_k=8, precision=4, scale_a=1e-6)
        run_gemm(1024, 272, 16385 // 4 * 4, 0, 0, split_k=22, precision=4)


def test_gemm_f16x2_without_operand_bounds_runs_the_exact_split():
    """A descriptor that asks for ADDHIP_PREC_F16X2 but carries no tracked maxima (or fused normalisation) runs as ADDHIP_PREC_BF16X3."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    rng = np.random.RandomState(3)
    A, B = T(rng.standard_normal((2048, 512)).astype(F) * 1e6), T(rng.standard_normal((1024, 512)).astype(F))
    outs = []
    for prec in (L.PREC_F16X2, L.PREC_BF16X3):
        Cm = torch.zeros(2048, 1024, device="cuda")
        L.call("addhip_gemm_f32", gemm(2048, 1024, 512, L.ptr(A), 512, 1, L.ptr(B), 512, 1, L.ptr(Cm), 1024, precision=prec), L.current_stream())
        outs.append(Cm)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())


@pytest.mark.parametrize("precision", [3, 2, 1])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_split_bf16_paths(a_kc, b_kc, precision):
    """The bf16-MFMA paths on shapes that take them (>= 256 tiles of 128x128): exact 3-way split within the fp32 bound,
    plain bf16 within the bf16 bound; ragged M/N/K edges, every layout, split-K, every epilogue."""
    run_gemm(16384 + (0 if not a_kc else 1), 1024, 264, a_kc, b_kc, precision=precision)
    run_gemm(4100, 1024, 1024 + 4, a_kc, b_kc, precision=precision)
    run_gemm(16384, 128, 512, a_kc, b_kc, precision=precision)  # 128 tiles: half a chip, still on the bf16-MFMA path
    if not a_kc and not b_kc:
        run_gemm(32, 512, 256, 0, 0, split_k=32, precision=precision)  # more split-K slices than 16-deep stages: empty slabs are zeros
    if not a_kc and not b_kc:
        run_gemm(1024, 264, 16385, 0, 0, split_k=22, precision=precision)
        run_gemm(1024, 1024, 16384, 0, 0, split_k=8, precision=precision)
    if a_kc:
        for epi in (1, 2, 3):
            run_gemm(16385, 512, 1024, 1, b_kc, epilogue=epi, precision=precision)
    if a_kc and b_kc:
        run_gemm(16384, 1024, 264, 1, 1, epilogue=2, norm=True, precision=precision)


@pytest.mark.parametrize("precision", [0, 3])
def test_gemm_split_k_accumulates_by_atomics(precision):
    """accumulate=1: the K slices add into ONE C instead of writing slabs (the weight-gradient GEMMs of the update step)."""
    run_gemm(1024, 1024, 16384, 0, 0, split_k=8, precision=precision, accumulate=1)
    run_gemm(1024, 272, 16385, 0, 0, split_k=22, precision=precision, accumulate=1)
    run_gemm(32, 512, 4096, 0, 0, split_k=32, precision=precision, accumulate=1)
    run_gemm(512, 1024, 16384, 0, 0, split_k=16, alpha=0.5, precision=precision, accumulate=1)


def test_gemm_split_error_is_at_fp32_level():
    """Measured worst error / sum|a||b| of the three product modes on one shape (the number DESIGN.md quotes)."""
    e32 = run_gemm(16384, 1024, 1024, 1, 1, precision=0)
    ex3 = run_gemm(16384, 1024, 1024, 1, 1, precision=3)
    ex2 = run_gemm(16384, 1024, 1024, 1, 1, precision=2)
    eb = run_gemm(16384, 1024, 1024, 1, 1, precision=1)
    print(f"worst |err| / sum|a||b|: fp32 MFMA {e32:.3e}, bf16x3 {ex3:.3e}, bf16x2 {ex2:.3e}, bf16 {eb:.3e}")
    assert ex3 <= 3.0 * e32 + 1e-8 and eb > 100 * ex3 and ex3 < ex2 < eb / 50


def test_gemm_fused_normalisation():
    run_gemm(777, 1024, 264, 1, 1, epilogue=2, norm=True)


@pytest.mark.parametrize("split", [2, 8, 16])
def test_gemm_split_k_weight_grad_shape(split):
    # dW[out,in] = dY^T X : both operands m-contiguous, reduction over the minibatch rows
    run_gemm(1024, 264, 4096 + 17, 0, 0, split_k=split)
    run_gemm(1024, 272, 4096 + 17, 0, 0, split_k=max(split, 11))  # >= 256 tiles of 128x96: the 96-wide configuration
    run_gemm(32, 512, 1000, 0, 0, split_k=split)


def test_gemm_single_row_and_alpha():
    run_gemm(1, 1024, 116, 1, 1, epilogue=2)
    run_gemm(1, 512, 1024, 1, 1, epilogue=2)
    for m in (1, 3, 8):  # the few-row kernel: both B layouts, every epilogue
        for epi in (0, 1, 2, 3):
            run_gemm(m, 1024, 512, 1, 0, epilogue=epi)
            run_gemm(m, 516, 128, 1, 1, epilogue=epi)
    run_gemm(200, 116, 1024, 1, 0, alpha=0.5)


def test_gemm_rejects_bad_arguments():
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    x = torch.zeros(64, 30, device="cuda")
    with pytest.raises(L.AddhipError):
        L.call("addhip_gemm_f32", gemm(64, 64, 30, L.ptr(x), 30, 1, L.ptr(x), 30, 1, L.ptr(x), 64), L.current_stream())
    with pytest.raises(L.AddhipError):
        L.call("addhip_gemm_f32", gemm(0, 64, 32, L.ptr(x), 32, 1, L.ptr(x), 32, 1, L.ptr(x), 64), L.current_stream())


def test_col_sum_and_slab_reduce():
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(1)
    X = rng.standard_normal((5001, 300)).astype(F)
    out = torch.full((300,), 5.0, device="cuda")
    L.call("addhip_col_sum", P(T(X)), 5001, 300, 300, L.ptr(out), 2.0, 0, L.current_stream())
    np.testing.assert_allclose(out.cpu().numpy(), 2.0 * X.astype(np.float64).sum(0), rtol=1e-5, atol=1e-3)
    L.call("addhip_col_sum", P(T(X)), 5001, 300, 300, L.ptr(out), 1.0, 1, L.current_stream())
    np.testing.assert_allclose(out.cpu().numpy(), 3.0 * X.astype(np.float64).sum(0), rtol=1e-5, atol=1e-3)
    S = rng.standard_normal((6, 1000)).astype(F)
    o2 = torch.ones(1000, device="cuda")
    L.call("addhip_slab_reduce", P(T(S)), 6, 1000, L.ptr(o2), 1000, 0.5, 1, L.current_stream())
    np.testing.assert_allclose(o2.cpu().numpy(), 1 + 0.5 * S.sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("precision", [0, 3])
@pytest.mark.parametrize("N", [1024, 272, 96])
def test_relu_sign_bits_roundtrip(N, precision):
    """BIAS_RELU writes 1 sign bit per element next to the activations; the MASK epilogue fed with those bits gives the
    same result, bit for bit, as with the fp32 activations as its mask (ragged M, N not a multiple of 32 or 128)."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    M, K = 16384 + 7, 128
    rng = np.random.RandomState(5)
    A, B, bias = rng.uniform(-1, 1, (M, K)).astype(F), rng.uniform(-1, 1, (N, K)).astype(F), rng.uniform(-1, 1, N).astype(F)
    dA, dB, dbias = T(A), T(B), T(bias)
    ldb = (N + 31) // 32
    H = torch.zeros(M, N, device="cuda")
    bits = torch.full((M, ldb), -1, dtype=torch.int32, device="cuda")
    st = L.current_stream()
    L.call("addhip_gemm_f32", gemm(M, N, K, L.ptr(dA), K, 1, L.ptr(dB), K, 1, L.ptr(H), N, 2, L.ptr(dbias), precision=precision,
                                   relu_bits=L.ptr(bits), ldbits=ldb), st)
    torch.cuda.synchronize()
    h = H.cpu().numpy()
    w = bits.cpu().numpy().astype(np.uint32)
    got = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(M, ldb * 32)
    assert np.array_equal(got[:, :N], (h > 0).astype(np.uint32)) and np.all(got[:, N:] == 0)
    assert 0.3 < got[:, :N].mean() < 0.7
    # backward through that ReLU: dX = (dY @ W) masked, W n-contiguous like the weights of the next layer
    K2 = 256
    dY, W2 = T(rng.uniform(-1, 1, (M, K2)).astype(F)), T(rng.uniform(-1, 1, (K2, N)).astype(F))
    out_f, out_b = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    cs_f, cs_b = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    L.call("addhip_gemm_f32", gemm(M, N, K2, L.ptr(dY), K2, 1, L.ptr(W2), N, 0, L.ptr(out_f), N, 3, mask=L.ptr(H), ldmask=N, colsum=L.ptr(cs_f),
                                   precision=precision), st)
    L.call("addhip_gemm_f32", gemm(M, N, K2, L.ptr(dY), K2, 1, L.ptr(W2), N, 0, L.ptr(out_b), N, 3, mask_bits=L.ptr(bits), ldbits=ldb,
                                   colsum=L.ptr(cs_b), precision=precision), st)
    torch.cuda.synchronize()
    assert torch.equal(out_b, out_f) and float((out_f != 0).float().mean()) > 0.3
    torch.testing.assert_close(cs_b, cs_f, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("N", [1024, 1000, 104])
def test_relu_sign_bits_roundtrip_bf16_storage(N):
    """The same round trip with bf16-stored operands and bf16 results (gemm_bf16_kernel's shared epilogue: sign-bit words assembled per
    row and stored by one lane per row; read back one word per row and handed out by readlane): bits == (h > 0) for the bf16 h that
    was stored, and the MASK epilogue gives the same dX from the bits as from an fp32 mask."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    M, K = 16384 + 9, 128
    rng = np.random.RandomState(6)
    bf = lambda a: torch.tensor(a).to(torch.bfloat16).cuda()
    dA, dB, dbias = bf(rng.uniform(-1, 1, (M, K)).astype(F)), bf(rng.uniform(-1, 1, (N, K)).astype(F)), T(rng.uniform(-1, 1, N).astype(F))
    ldb = (N + 31) // 32
    ld16 = (N + 7) // 8 * 8
    H = torch.zeros(M, N, device="cuda")
    H16 = torch.zeros(M, ld16, device="cuda", dtype=torch.bfloat16)
    bits = torch.full((M, ldb), -1, dtype=torch.int32, device="cuda")
    st = L.current_stream()
    k16 = dict(precision=L.PREC_BF16, operands_bf16=1)
    L.call("addhip_gemm_f32", gemm(M, N, K, L.ptr(dA), K, 1, L.ptr(dB), K, 1, L.ptr(H), N, 2, L.ptr(dbias), relu_bits=L.ptr(bits), ldbits=ldb,
                                   C16=L.ptr(H16), ldc16=ld16, **k16), st)
    torch.cuda.synchronize()
    h = H.cpu().numpy()
    w = bits.cpu().numpy().astype(np.uint32)
    got = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(M, ldb * 32)
    assert np.array_equal(got[:, :N], (h > 0).astype(np.uint32)) and np.all(got[:, N:] == 0)
    assert torch.equal(H16[:, :N], H.to(torch.bfloat16)) and 0.3 < got[:, :N].mean() < 0.7
    K2 = 256
    dY, W2t = bf(rng.uniform(-1, 1, (M, K2)).astype(F)), bf(rng.uniform(-1, 1, (N, K2)).astype(F))  # W^T [in, out]: k-contiguous
    out_f, out_b = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    out16 = torch.zeros(M, ld16, device="cuda", dtype=torch.bfloat16)
    cs_f, cs_b = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    L.call("addhip_gemm_f32", gemm(M, N, K2, L.ptr(dY), K2, 1, L.ptr(W2t), K2, 1, L.ptr(out_f), N, 3, mask=L.ptr(H), ldmask=N, colsum=L.ptr(cs_f), **k16), st)
    L.call("addhip_gemm_f32", gemm(M, N, K2, L.ptr(dY), K2, 1, L.ptr(W2t), K2, 1, L.ptr(out_b), N, 3, mask_bits=L.ptr(bits), ldbits=ldb,
                                   colsum=L.ptr(cs_b), C16=L.ptr(out16), ldc16=ld16, **k16), st)
    torch.cuda.synchronize()
    assert torch.equal(out_b, out_f) and float((out_f != 0).float().mean()) > 0.3
    assert torch.equal(out16[:, :N], out_b.to(torch.bfloat16))
    torch.testing.assert_close(cs_b, cs_f, rtol=1e-4, atol=1e-3)


def run_gemm_bf16(M, N, K, a_kc, b_kc, epilogue=0, split_k=1, seed=0, both_outputs=True, hint=0):
    """Operands STORED as bf16 (addhip_gemm_t.operands_bf16): the products are exact, so against float64 on the same bf16 values the
    only error is the fp32 accumulation; the optional bf16 result copy must be the fp32 result rounded to nearest even."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    rng = np.random.RandomState(seed)
    A = torch.tensor(rng.uniform(-1, 1, (M, K)).astype(F)).to(torch.bfloat16)
    B = torch.tensor(rng.uniform(-1, 1, (N, K)).astype(F)).to(torch.bfloat16)
    bias = rng.uniform(-1, 1, N).astype(F)
    mask = rng.uniform(-1, 1, (M, N)).astype(F)
    dA = (A if a_kc else A.t().contiguous()).cuda()
    dB = (B if b_kc else B.t().contiguous()).cuda()
    lda, ldb = (K if a_kc else M), (K if b_kc else N)
    ldc = (N + 7) // 8 * 8
    dC = torch.full((max(split_k, 1), M, ldc), 9.0, device="cuda")
    dC16 = torch.full((M, ldc), 9.0, device="cuda", dtype=torch.bfloat16)
    dbias, dmask = T(bias), T(mask)
    dcs = torch.full((N,), 0.5, device="cuda")
    use16 = both_outputs and split_k <= 1
    g = gemm(M, N, K, L.ptr(dA), lda, a_kc, L.ptr(dB), ldb, b_kc, L.ptr(dC), ldc, epilogue, L.ptr(dbias), L.ptr(dmask), N, None, None, split_k, 1.0,
             L.ptr(dcs) if epilogue == 3 else None, 1, operands_bf16=1, C16=L.ptr(dC16) if use16 else None, ldc16=ldc, hint=hint)
    L.call("addhip_gemm_f32", g, L.current_stream())
    torch.cuda.synchronize()
    A64, B64 = A.double().numpy(), B.double().numpy()
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
    assert np.all(err <= 4e-7 * scale + 1e-6), (M, N, K, a_kc, b_kc, epilogue, split_k, float((err / np.maximum(scale, 1e-30)).max()))
    if ldc > N:
        assert np.all(out[0][:, N:] == 9.0)
    if use16:
        want16 = torch.tensor(dC[0].cpu().numpy()[:, :N]).to(torch.bfloat16)
        assert torch.equal(dC16.cpu()[:, :N], want16)  # same fp32 value, rounded to nearest even
    if epilogue == 3:
        cs = dcs.cpu().numpy().astype(np.float64) - 0.5
        cs_scale = np.where(mask > 0, scale, 0).sum(0)
        assert np.all(np.abs(cs - got.sum(0)) <= 4e-7 * cs_scale + 1e-5)


@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0), (0, 1)])
def test_gemm_bf16_storage(a_kc, b_kc):
    """The bf16-storage GEMM (gemm_bf16.hip): every operand layout, ragged edges, K tails, epilogues, split-K slabs."""
    M0 = 4104 if not a_kc else 4100  # m-contiguous operands move 8 rows per load
    N0 = 1032 if not b_kc else 1000
    run_gemm_bf16(M0, N0, 1024 + 8, a_kc, b_kc)
    run_gemm_bf16(256, 128, 64, a_kc, b_kc)
    run_gemm_bf16(16384 if not a_kc else 16385, 512, 272, a_kc, b_kc)
    if a_kc:
        for epi in (1, 2, 3):
            run_gemm_bf16(1000, 512, 1024, 1, b_kc, epilogue=epi)
    if not a_kc and not b_kc:  # the weight-gradient shapes: K = minibatch rows (odd count: the discriminator's Mb + 1), split-K slabs
        run_gemm_bf16(1024, 272, 16385, 0, 0, split_k=22)
        run_gemm_bf16(1024, 1024, 16384, 0, 0, split_k=8)
        run_gemm_bf16(32, 512, 4096, 0, 0, split_k=32)


@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0), (0, 1)])
def test_gemm_bf16_storage_256_tiles(a_kc, b_kc):
    """The 256x256-tile kernel (gemm_bf16_q_kernel: unit ring, counted vmcnt, quadrant phases), forced on for shapes the
    dispatcher would give to the 128x128 kernel: every operand layout, every epilogue, split-K, one and several K tiles."""
    big = 1  # ADDHIP_GEMM_HINT_BIG_TILE
    run_gemm_bf16(512, 768, 64, a_kc, b_kc, hint=big)
    run_gemm_bf16(256, 256, 1024 + 64, a_kc, b_kc, hint=big)
    if a_kc:
        for epi in (1, 2, 3):
            run_gemm_bf16(512, 256, 448, 1, b_kc, epilogue=epi, hint=big)
    run_gemm_bf16(256, 512, 4096, a_kc, b_kc, split_k=6, hint=big)


@pytest.mark.parametrize("precision", ["fp32", "bf16x2"])
def test_gemm_fp32_operands_write_a_bf16_copy(precision):
    """C16 with fp32 operands (what the bf16-storage plan uses for the actor head's dX): the fp32 result rounded to nearest even,
    with and without the fp32 C, also for a few-row launch (which must not take the fp32-only few-row kernel)."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    prec = {"fp32": L.PREC_F32, "bf16x2": L.PREC_BF16X2}[precision]
    rng = np.random.RandomState(5)
    for M, N, K in ((4100, 512, 32), (5, 512, 64)):
        A, B = T(rng.uniform(-1, 1, (M, K)).astype(F)), T(rng.uniform(-1, 1, (N, K)).astype(F))
        C = torch.zeros(M, N, device="cuda")
        C16 = torch.full((M, N), 9.0, device="cuda", dtype=torch.bfloat16)
        C16b = torch.full((M, N), 9.0, device="cuda", dtype=torch.bfloat16)
        g = gemm(M, N, K, L.ptr(A), K, 1, L.ptr(B), K, 1, L.ptr(C), N, precision=prec, C16=L.ptr(C16), ldc16=N)
        L.call("addhip_gemm_f32", g, L.current_stream())
        g2 = gemm(M, N, K, L.ptr(A), K, 1, L.ptr(B), K, 1, None, N, precision=prec, C16=L.ptr(C16b), ldc16=N)
        L.call("addhip_gemm_f32", g2, L.current_stream())
        torch.cuda.synchronize()
        ref = A.double() @ B.double().t()
        assert float((C.double() - ref).abs().max()) < (1e-5 if precision == "fp32" else 2e-3)
        assert torch.equal(C16, C.to(torch.bfloat16)) and torch.equal(C16b, C16)


def test_to_bf16_rounds_to_nearest_even():
    import torch
    import add_gym_amd._lib as L

    rng = np.random.RandomState(1)
    x = np.concatenate([rng.standard_normal(4096 * 12).astype(F) * 10, np.asarray([1.00390625, 1.01171875, -1.00390625, 3.0e-39], F)]).reshape(-1, 4)
    src = torch.zeros(x.shape[0], 8, device="cuda")
    src[:, :4] = T(x)
    dst = torch.full((x.shape[0], 12), 7.0, device="cuda", dtype=torch.bfloat16)
    L.call("addhip_to_bf16", L.ptr(src), L.ptr(dst), x.shape[0], 4, 8, 12, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :4].cpu(), torch.tensor(x).to(torch.bfloat16)) and bool((dst[:, 4:] == 7.0).all())


def test_normalize_to_bf16_is_the_normaliser_then_round_to_nearest_even():
    """addhip_normalize_to_bf16: (x - mean) / std in fp32 (Normalizer.normalize, normalizer.py:107-110), then bf16 round to nearest even --
    bit for bit what torch gives for the same two steps; columns past `cols` of the destination rows are left alone."""
    import torch
    import add_gym_amd._lib as L

    torch.manual_seed(5)
    rows, cols, ld_src, ld_dst = 777, 272, 272, 280
    src = torch.randn(rows, ld_src, device="cuda") * 3
    mean, std = torch.randn(cols, device="cuda"), torch.rand(cols, device="cuda") + 0.2
    dst = torch.full((rows, ld_dst), 7.0, device="cuda", dtype=torch.bfloat16)
    L.call("addhip_normalize_to_bf16", L.ptr(src), L.ptr(mean), L.ptr(std), L.ptr(dst), rows, cols, ld_src, ld_dst, L.current_stream())
    torch.cuda.synchronize()
    want = ((src[:, :cols] - mean) / std).to(torch.bfloat16)
    assert torch.equal(dst[:, :cols], want) and bool((dst[:, cols:] == 7.0).all())


def test_shadow_refresh_flat_and_transposed_in_one_launch():
    """addhip_shadow_refresh == addhip_to_bf16 over the flat buffer + addhip_to_bf16_t per listed matrix (ragged 32x32 tiles included)."""
    import ctypes as C
    import torch
    import add_gym_amd._lib as L

    torch.manual_seed(3)
    mats = [(0, 1024, 272), (1024 * 272 + 1024, 512, 1024), (1024 * 272 + 1024 + 512 * 1024 + 512, 40, 100)]
    count = mats[-1][0] + 40 * 100 + 7
    params = torch.randn(count, device="cuda")
    flat16 = torch.full((count,), 9.0, device="cuda", dtype=torch.bfloat16)
    trans16 = torch.full((count,), 9.0, device="cuda", dtype=torch.bfloat16)
    n = len(mats)
    L.call("addhip_shadow_refresh", L.ptr(params), L.ptr(flat16), L.ptr(trans16), count, (C.c_int64 * n)(*(m[0] for m in mats)),
           (C.c_int32 * n)(*(m[1] for m in mats)), (C.c_int32 * n)(*(m[2] for m in mats)), n, L.STORE_BF16, L.current_stream())
    torch.cuda.synchronize()
    assert torch.equal(flat16, params.to(torch.bfloat16))
    covered = torch.zeros(count, dtype=torch.bool, device="cuda")
    for off, r, c in mats:
        want = params[off:off + r * c].view(r, c).t().contiguous().to(torch.bfloat16)
        assert torch.equal(trans16[off:off + r * c].view(c, r), want)
        covered[off:off + r * c] = True
    assert bool((trans16[~covered] == 9.0).all())
    with pytest.raises(RuntimeError):
        L.call("addhip_shadow_refresh", L.ptr(params), L.ptr(flat16), L.ptr(trans16), count, (C.c_int64 * 1)(count - 10), (C.c_int32 * 1)(8),
               (C.c_int32 * 1)(8), 1, L.STORE_BF16, L.current_stream())


def _run_with_hint(hint, *a, **k):
    """run_gemm with addhip_gemm_t.hint set on the descriptor it builds."""
    import add_gym_amd.hotpath as H

    orig = H.gemm
    H.gemm = lambda *aa, **kk: orig(*aa, **dict(kk, hint=hint))
    try:
        return run_gemm(*a, **k)
    finally:
        H.gemm = orig


@pytest.mark.parametrize("stages", [4, 8, 16])  # ADDHIP_GEMM_HINT_ONE_STAGE / TWO_STAGE (LDS-DMA kernel), REG_STAGED (the register-staged kernel)
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_gemm_fp32_lds_dma_kernel(a_kc, b_kc, stages):
    """fp32 operands on the 128x128 LDS-DMA kernel (gemm_dma.h; what shapes of more than 256 tiles take): every layout, ragged M / N / K
    edges (K tail inside a 32-deep stage, rows past the last tile, N not a multiple of 128), epilogues with sign-bit masks' fp32
    fallback, split-K slabs with empty and ragged slices -- under both stage configurations and, for comparison, the register-staged
    kernel on the same shapes."""
    run = lambda *a, **k: _run_with_hint(stages, *a, **k)
    run(8192 + 128 + 4, 1024, 32 - 4, a_kc, b_kc)   # one (ragged) stage per tile
    run(16384 + (0 if not a_kc else 1), 1024, 264, a_kc, b_kc)
    run(4100, 1024, 1024 + 4, a_kc, b_kc)
    run(2048 + 4, 2048 - 4 * 9, 96 + 4, a_kc, b_kc)
    if not a_kc and not b_kc:
        run(1024, 512, 16385, 0, 0, split_k=22)
        run(1024, 1024, 16384, 0, 0, split_k=8)
        run(512, 1024, 300, 0, 0, split_k=16)  # slices shorter than a stage, some empty
    if a_kc:
        for epi in (1, 2, 3):
            run(16385, 512, 1024, 1, b_kc, epilogue=epi)
        run(8200, 1024, 512, 1, b_kc, alpha=0.5)


@pytest.mark.parametrize("bf16", [0, 1])
@pytest.mark.parametrize("count", [2, 3])
def test_gemm_grouped_equals_single_launches(count, bf16):
    """addhip_gemm_grouped: `count` equal-shaped problems on different buffers in one launch == the same problems launched one by one,
    bit for bit (fp32 and bf16-stored operands; forward, dX and split-K weight-gradient shapes; a shape that is launched one by one)."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    rng = np.random.RandomState(11)
    dt = torch.bfloat16 if bf16 else torch.float32
    esz = 2 if bf16 else 4
    for (M, N, K, akc, bkc, epi, split) in ((4100, 1024, 264 if not bf16 else 272, 1, 1, 2, 1), (4096, 512, 1024, 1, 0, 3, 1), (1024, 1024, 4100 if not bf16 else 4104, 0, 0, 0, 8),
                                            (256, 64, 128, 1, 1, 1, 1)):
        probs, outs = [], []
        for i in range(count):
            A = torch.tensor(rng.uniform(-1, 1, (M, K) if akc else (K, M)).astype(F)).to(dt).cuda()
            B = torch.tensor(rng.uniform(-1, 1, (N, K) if bkc else (K, N)).astype(F)).to(dt).cuda()
            bias, mask = T(rng.uniform(-1, 1, N).astype(F)), T(rng.uniform(-1, 1, (M, N)).astype(F))
            C1, C2 = (torch.full((max(split, 1), M, N), 7.0, device="cuda") for _ in range(2))
            cs1, cs2 = (torch.zeros(N, device="cuda") for _ in range(2))
            mk = lambda C, cs: gemm(M, N, K, P(A), K if akc else M, akc, P(B), K if bkc else N, bkc, P(C), N, epi, P(bias), P(mask), N, split_k=split,
                                    colsum=P(cs) if epi == 3 else None, precision=L.PREC_BF16 if bf16 else L.PREC_F32, operands_bf16=bf16)
            probs.append((mk(C1, cs1), mk(C2, cs2)))
            outs.append((C1, C2, cs1, cs2))
        arr = (L.GemmT * count)(*[p[0] for p in probs])
        L.call("addhip_gemm_grouped", arr, count, L.current_stream())
        for p in probs:
            L.call("addhip_gemm_f32", p[1], L.current_stream())
        torch.cuda.synchronize()
        for C1, C2, cs1, cs2 in outs:
            assert torch.equal(C1, C2) and not bool((C1 == 7.0).all()), (M, N, K)
            torch.testing.assert_close(cs1, cs2, rtol=1e-4, atol=2e-3)  # column sums of 4096 values go by float atomics: order varies
    # mismatched problems are refused
    g1 = gemm(512, 512, 512, P(torch.zeros(512 * 512, device="cuda")), 512, 1, P(torch.zeros(512 * 512, device="cuda")), 512, 1, P(torch.zeros(512 * 512, device="cuda")), 512)
    g2 = gemm(512, 512, 256, g1.A, 512, 1, g1.B, 512, 1, g1.C, 512)
    with pytest.raises(L.AddhipError):
        L.call("addhip_gemm_grouped", (L.GemmT * 2)(g1, g2), 2, L.current_stream())


@pytest.mark.parametrize("hint", [4, 8])  # ADDHIP_GEMM_HINT_ONE_STAGE / TWO_STAGE
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 0), (0, 1)])
def test_gemm_bf16_storage_stage_configurations(a_kc, b_kc, hint):
    """bf16-stored operands under both stage configurations of the 128x128 LDS-DMA kernel, whatever the dispatcher would pick: ragged rows
    past the last tile, K tails inside a 64-deep stage, 1 / 5 / 16 stages per tile, every epilogue, split-K slabs."""
    run_gemm_bf16(16384 + (8 if not a_kc else 1), 1024, 272, a_kc, b_kc, hint=hint)
    run_gemm_bf16(8192 + 128, 1024, 56, a_kc, b_kc, hint=hint)
    if a_kc:
        for epi in (1, 2, 3):
            run_gemm_bf16(16385, 512, 1024, 1, b_kc, epilogue=epi, hint=hint)
    if not a_kc and not b_kc:
        run_gemm_bf16(1024, 272, 16385 + 7, 0, 0, split_k=22, hint=hint)
        run_gemm_bf16(1024, 1024, 16384, 0, 0, split_k=8, hint=hint)


@pytest.mark.parametrize("bf16", [False, True])
def test_replicated_column_sums_and_the_paired_combine(bf16):
    """addhip_gemm_t.colsum_replicas: the MASK epilogue's bias-gradient column sums spread over 16 rows (block index % 16) add up to the plain
    colsum; addhip_slab_reduce_pair sums the rows into the gradient, clears them, and does the split-K combine of its first job exactly
    like addhip_slab_reduce (ragged M: the last row tile takes the general epilogue, the others the straight-line one)."""
    import torch
    import add_gym_amd._lib as L
    from add_gym_amd.hotpath import gemm

    M, N, K, R = 16384 + 40, 512, 128, 16
    rng = np.random.RandomState(11)
    A, W = rng.uniform(-1, 1, (M, K)).astype(F), rng.uniform(-1, 1, (N, K)).astype(F)
    bits = torch.tensor(rng.randint(-2**31, 2**31 - 1, (M, N // 32)), dtype=torch.int32, device="cuda")
    if bf16:
        dA, dW = torch.tensor(A).to(torch.bfloat16).cuda(), torch.tensor(W).to(torch.bfloat16).cuda()
        kw = dict(precision=L.PREC_BF16, operands_bf16=1)
    else:
        dA, dW, kw = T(A), T(W), {}
    st = L.current_stream()
    out1, out2 = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    cs_plain, reps = torch.zeros(N, device="cuda"), torch.zeros(R, N, device="cuda")
    L.call("addhip_gemm_f32", gemm(M, N, K, L.ptr(dA), K, 1, L.ptr(dW), K, 1, L.ptr(out1), N, 3, mask_bits=L.ptr(bits), ldbits=N // 32, colsum=L.ptr(cs_plain), **kw), st)
    L.call("addhip_gemm_f32", gemm(M, N, K, L.ptr(dA), K, 1, L.ptr(dW), K, 1, L.ptr(out2), N, 3, mask_bits=L.ptr(bits), ldbits=N // 32, colsum=L.ptr(reps),
                                   colsum_replicas=R, ldcs=N, **kw), st)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2)
    want = out1.double().sum(0)
    assert float(reps.abs().min(1).values.max()) > 0  # every replica row received sums
    torch.testing.assert_close(reps.double().sum(0), want, rtol=1e-5, atol=1e-2)
    torch.testing.assert_close(cs_plain.double(), want, rtol=1e-5, atol=1e-2)
    # row r of the replicas = the column sums of the 32-row blocks with index % 16 == r
    blk = torch.arange(M, device="cuda") // 32 % R
    for r in (0, 5, 15):
        torch.testing.assert_close(reps[r].double(), out1[blk == r].double().sum(0), rtol=1e-5, atol=1e-2)
    # the paired combine: job 1 == addhip_slab_reduce, job 2 sums and clears the replica rows
    slabs = T(rng.standard_normal((6, 4096)).astype(F))
    o_ref, o_pair = torch.ones(4096, device="cuda"), torch.ones(4096, device="cuda")
    gb = torch.full((N,), 2.0, device="cuda")
    L.call("addhip_slab_reduce", L.ptr(slabs), 6, 4096, L.ptr(o_ref), 4096, 0.5, 1, st)
    L.call("addhip_slab_reduce_pair", L.ptr(slabs), 6, 4096, L.ptr(o_pair), 4096, 0.5, 1, L.ptr(reps), R, N, L.ptr(gb), N, 1, 1, st)
    torch.cuda.synchronize()
    assert torch.equal(o_pair, o_ref)
    torch.testing.assert_close(gb.double(), 2.0 + want, rtol=1e-5, atol=1e-2)
    assert float(reps.abs().max()) == 0.0
    lib = L.load()
    assert lib.addhip_gemm_f32(gemm(M, N, K, L.ptr(dA), K, 1, L.ptr(dW), K, 1, L.ptr(out2), N, 3, mask_bits=L.ptr(bits), ldbits=N // 32, colsum=L.ptr(reps),
                                    colsum_replicas=R, ldcs=N - 4, **kw), st) != 0  # rows of the replicas shorter than N

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.hotpath import gemm
F = np.float32
def run(M, N, K, epi, bf16, hint):
    rng = np.random.RandomState(0)
    dt = torch.bfloat16 if bf16 else torch.float32
    A = torch.tensor(rng.uniform(-1, 1, (M, K)).astype(F)).to(dt)
    B = torch.tensor(rng.uniform(-1, 1, (N, K)).astype(F)).to(dt)
    bias = torch.tensor(rng.uniform(-1, 1, N).astype(F)).cuda()
    mask = torch.tensor(rng.uniform(-1, 1, (M, N)).astype(F)).cuda()
    dA, dB = A.cuda(), B.cuda()
    C = torch.full((M, N), 9.0, device="cuda")
    g = gemm(M, N, K, L.ptr(dA), K, 1, L.ptr(dB), K, 1, L.ptr(C), N, epi, L.ptr(bias), L.ptr(mask), N, precision=L.PREC_BF16 if bf16 else 0, operands_bf16=int(bf16), hint=hint)
    L.call("addhip_gemm_f32", g, L.current_stream()); torch.cuda.synchronize()
    ref = A.double() @ B.double().T
    if epi in (1, 2): ref = ref + bias.cpu().double()
    if epi == 2: ref = ref.clamp(min=0)
    if epi == 3: ref = torch.where(mask.cpu() > 0, ref, torch.zeros_like(ref))
    err = (C.cpu().double() - ref).abs()
    bad = (err > 1e-3).nonzero()
    print(f"M={M} N={N} K={K} epi={epi} bf16={bf16} hint={hint}: bad={len(bad)}", "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:12], "...", sorted(set(bad[:, 1].tolist()))[-4:] if len(bad) else "", flush=True)
for bf16 in (1, 0):
    for epi in (0, 1, 2, 3):
        run(16385, 512, 1024, epi, bf16, 32)
    run(16384, 512, 1024, 1, bf16, 32)
    run(16384 * 2, 512, 1024, 1, bf16, 32)
    run(16384 * 2, 512, 1024, 2, bf16, 32)
    run(16385, 512, 1024, 1, bf16, 64)

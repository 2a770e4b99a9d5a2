"""One bf16-storage GEMM shape launched a few times (for rocprofv3 PMC passes).  usage: gemm_bf16_one.py [M N K] [akc bkc epi] [split] [hint]  (hint: ADDHIP_GEMM_HINT_* bits, 1 = 256x256 kernel, 2 = never)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.hotpath import gemm
a = [int(x) for x in sys.argv[1:]]
M, N, K = a[0:3] if len(a) >= 3 else (16384, 1024, 1024)
akc, bkc, epi = a[3:6] if len(a) >= 6 else (1, 1, 2)
split = a[6] if len(a) >= 7 else 1
hint = a[7] if len(a) >= 8 else 0
bf = lambda n: (torch.zeros(n, device="cuda") if os.environ.get("ZERO") else torch.randn(n, device="cuda")).to(torch.bfloat16)
A, B, bias = bf(M * K), bf(N * K), torch.randn(N, device="cuda")
C = torch.zeros(M * N * split, device="cuda") if split > 1 else None
C16 = None if split > 1 else torch.zeros(M * N, device="cuda", dtype=torch.bfloat16)
bits = torch.randint(-2**31, 2**31 - 1, (M * ((N + 31) // 32),), device="cuda", dtype=torch.int32)
kw = dict(mask_bits=L.ptr(bits), ldbits=(N + 31) // 32) if epi == 3 else (dict(relu_bits=L.ptr(bits), ldbits=(N + 31) // 32) if epi == 2 else {})
g = gemm(M, N, K, L.ptr(A), K if akc else M, akc, L.ptr(B), K if bkc else N, bkc, L.ptr(C) if C is not None else None, N, epi, L.ptr(bias), None, 0,
         precision=L.PREC_BF16, split_k=split, operands_bf16=1, C16=L.ptr(C16) if C16 is not None else None, ldc16=N, hint=hint, **kw)
st = torch.cuda.current_stream()
for _ in range(3):
    L.call("addhip_gemm_f32", g, st.cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(10):
    L.call("addhip_gemm_f32", g, st.cuda_stream)
e1.record(st); e1.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"bf16 storage M={M} N={N} K={K} akc={akc} bkc={bkc} epi={epi} split={split}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)

"""One GEMM shape launched a few times (for rocprofv3 PMC passes).  usage: gemm_one.py [precision 0|1|2|3|4] [M N K] [akc bkc epi split]
(4 = ADDHIP_PREC_F16X2: the operands' tracked maxima are computed first with addhip_amax_f32)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.hotpath import gemm
a = [int(x) for x in sys.argv[1:]]
prec = a[0] if a else 0
M, N, K = a[1:4] if len(a) >= 4 else (16384, 1024, 1024)
akc, bkc, epi = a[4:7] if len(a) >= 7 else (1, 1, 2)
split = a[7] if len(a) >= 8 else 1
A, B, C, bias = torch.randn(M*K, device="cuda"), torch.randn(N*K, device="cuda"), torch.zeros(M*N*split, device="cuda"), torch.randn(N, device="cuda")
mask = torch.randn(M*N, device="cuda")
g = gemm(M, N, K, L.ptr(A), K if akc else M, akc, L.ptr(B), K if bkc else N, bkc, L.ptr(C), N, epi, L.ptr(bias), L.ptr(mask), (0 if os.environ.get('MASK_LD0') else N), precision=prec, split_k=split, hint=int(os.environ.get("HINT", "0")))
st = torch.cuda.current_stream()
if prec == 4:
    am = torch.zeros(2, L.AMAX_SLOTS, dtype=torch.int32, device="cuda")
    L.call("addhip_amax_f32", L.ptr(A), A.numel(), L.ptr(am[0]), st.cuda_stream)
    L.call("addhip_amax_f32", L.ptr(B), B.numel(), L.ptr(am[1]), st.cuda_stream)
    g.a_amax, g.b_amax = L.ptr(am[0]), L.ptr(am[1])
_w = torch.randn(8192, 8192, device="cuda")
for _ in range(30): _w @ _w
for _ in range(3):
    L.call("addhip_gemm_f32", g, st.cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(5):
    L.call("addhip_gemm_f32", g, st.cuda_stream)
e1.record(st); e1.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"prec={prec} M={M} N={N} K={K} akc={akc} bkc={bkc} epi={epi}: {ms*1e3:.1f} us  {2.0*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)

"""One plane-storage GEMM shape (csrc/gemm_x3.hip) timed with HIP events: gemm_x3_one.py M N K [akc bkc epi split hint] ; hint 1 / 32 / 2 =
256x256 / 256x128 / 128x128 tiles (0: the dispatcher's choice).  Random bf16 values in every plane (zeros would flatter the clock)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.hotpath import gemm
a = [int(x) for x in sys.argv[1:]] + [0] * 8
M, N, K, akc, bkc, epi, split, hint = a[:8]
if len(sys.argv) < 5: akc, bkc = 1, 1
split = max(split, 1)
dev = "cuda"
A = torch.randn(M * K * 3, device=dev).to(torch.bfloat16)
B = torch.randn(N * K * 3, device=dev).to(torch.bfloat16)
C = torch.zeros(split, M, N, device=dev)
C16 = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
bias = torch.randn(N, device=dev)
bits = torch.randint(0, 2 ** 31 - 1, (M, (N + 31) // 32), dtype=torch.int32, device=dev)
cs = torch.zeros(16, N, device=dev)
g = gemm(M, N, K, L.ptr(A), K if akc else M, akc, L.ptr(B), K if bkc else N, bkc, L.ptr(C) if split > 1 or epi == 0 else None, N, epi, L.ptr(bias), None, 0, None, None,
         split, 1.0, L.ptr(cs) if epi == 3 else None, L.PREC_BF16X3, relu_bits=L.ptr(bits) if epi == 2 else None, mask_bits=L.ptr(bits) if epi == 3 else None,
         ldbits=(N + 31) // 32, operands_bf16=3, C16=L.ptr(C16) if split == 1 and epi != 0 else None, ldc16=N, hint=hint, colsum_replicas=16 if epi == 3 else 0, ldcs=N,
         c16_planes=3)
st = torch.cuda.current_stream()
_w = torch.randn(8192, 8192, device=dev)
for _ in range(40): _w @ _w
for _ in range(5): L.call("addhip_gemm_f32", g, st.cuda_stream)
torch.cuda.synchronize()
ts = []
for rep in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10): L.call("addhip_gemm_f32", g, st.cuda_stream)
    e1.record(st); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
ms = sorted(ts)[4]
print(f"M={M} N={N} K={K} akc={akc} bkc={bkc} epi={epi} split={split} hint={hint}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF (fp32-equivalent; x6 bf16 MFMA work)", flush=True)

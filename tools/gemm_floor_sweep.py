"""Where a short-K bf16-storage GEMM launch spends its time: M=16384, N=1024 at K = 64 ... 1024 under each epilogue, 10 back-to-back launches per
point (HIP events), plus an empty-kernel-sized reference (addhip_fill_zero of 4 floats) for the per-launch overhead of the measurement.
Fit: time = floor(epilogue) + K * slope.  usage: gemm_floor_sweep.py [M N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.hotpath import gemm
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 1024)
st = torch.cuda.current_stream()
_w = torch.randn(8192, 8192, device="cuda")
for _ in range(40): _w @ _w
torch.cuda.synchronize()

def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

z = torch.zeros(4, device="cuda")
print(f"addhip_fill_zero(4 floats), back to back: {timed(lambda: L.call('addhip_fill_zero', L.ptr(z), 4, st.cuda_stream)):.1f} us per launch")
big = torch.zeros(M * N, device="cuda", dtype=torch.bfloat16)
print(f"torch fill of the {M*N*2/1e6:.0f} MB bf16 result: {timed(lambda: big.fill_(1.0)):.1f} us")
bias = torch.randn(N, device="cuda")
bits = torch.randint(-2**31, 2**31 - 1, (M * ((N + 31) // 32),), device="cuda", dtype=torch.int32)
C16 = torch.zeros(M * N, device="cuda", dtype=torch.bfloat16)
C32 = torch.zeros(M * N, device="cuda")
for epi, name in ((0, "none"), (2, "bias+relu+bits"), (3, "mask(bits)+colsum"), (-3, "mask(bits)")):
    for out in ("bf16", "fp32"):
        row = []
        for K in (64, 128, 256, 512, 1024):
            A, B = torch.randn(M * K, device="cuda").to(torch.bfloat16), torch.randn(N * K, device="cuda").to(torch.bfloat16)
            cs = torch.zeros(N, device="cuda")
            kw = dict(mask_bits=L.ptr(bits), ldbits=(N + 31) // 32, **(dict(colsum=L.ptr(cs)) if epi == 3 else {})) if abs(epi) == 3 else (dict(relu_bits=L.ptr(bits), ldbits=(N + 31) // 32) if epi == 2 else {})
            g = gemm(M, N, K, L.ptr(A), K, 1, L.ptr(B), K, 1, L.ptr(C32) if out == "fp32" else None, N, abs(epi), L.ptr(bias), precision=L.PREC_BF16, operands_bf16=1,
                     C16=L.ptr(C16) if out == "bf16" else None, ldc16=N, **kw)
            row.append(timed(lambda: L.call("addhip_gemm_f32", g, st.cuda_stream)))
        print(f"epilogue {name:18s} out {out}: K=64..1024 " + " ".join(f"{t:6.1f}" for t in row) + f" us;  floor ~{2*row[0]-row[1]:.1f} us, +{(row[4]-row[2])/768*64:.2f} us per 64-deep stage", flush=True)

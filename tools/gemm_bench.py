"""Microbenchmark of addhip_gemm_f32 on the shapes of one optimiser step (HIP events on the launch stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.hotpath import gemm

SHAPES = [  # (name, M, N, K, a_kc, b_kc, split, epilogue)
    ("fwd L1  16384x1024x264", 16384, 1024, 264, 1, 1, 1, 2),
    ("fwd L2  16384x1024x1024", 16384, 1024, 1024, 1, 1, 1, 2),
    ("fwd L3  16384x512x1024", 16384, 512, 1024, 1, 1, 1, 2),
    ("dX  L3  16384x1024x512", 16384, 1024, 512, 1, 0, 1, 3),
    ("dX  L2  16384x1024x1024", 16384, 1024, 1024, 1, 0, 1, 3),
    ("dW  L2  1024x1024x16384 s8", 1024, 1024, 16384, 0, 0, 8, 0),
    ("dW  L3  512x1024x16384 s16", 512, 1024, 16384, 0, 0, 16, 0),
    ("dW  L1  1024x264x16384 s22", 1024, 264, 16384, 0, 0, 22, 0),
    ("roll L2 4096x1024x1024", 4096, 1024, 1024, 1, 1, 1, 2),
    ("roll L3 4096x512x1024", 4096, 512, 1024, 1, 1, 1, 2),
]
dev = "cuda"
big = torch.randn(64 * 1024 * 1024, device=dev)
buf = (lambda n: torch.zeros(n, device=dev)) if os.environ.get("ZERO") else (lambda n: torch.randn(n, device=dev))
tot_f, tot_t = 0.0, 0.0
for name, M, N, K, akc, bkc, split, epi in SHAPES:
    A, B = buf(M * K), buf(N * K)
    C = torch.zeros(max(split, 1) * M * N, device=dev)
    bias, mask = buf(N), buf(M * N)
    g = gemm(M, N, K, L.ptr(A), K if akc else M, akc, L.ptr(B), K if bkc else N, bkc, L.ptr(C), N, epi, L.ptr(bias), L.ptr(mask), N, split_k=split)
    st = torch.cuda.current_stream()
    for _ in range(3):
        L.call("addhip_gemm_f32", g, st.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(st)
    for _ in range(reps):
        L.call("addhip_gemm_f32", g, st.cuda_stream)
    e1.record(st); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * M * N * K
    tot_f += fl; tot_t += ms
    print(f"{name:30s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s", flush=True)
print(f"{'weighted total':30s} {tot_t*1e3:8.1f} us  {tot_f/tot_t/1e9:7.1f} TFLOP/s")

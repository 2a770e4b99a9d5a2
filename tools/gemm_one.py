import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, add_gym_amd
import add_gym_amd._lib as L
from add_gym_amd.hotpath import gemm
M, N, K = 16384, 1024, 1024
A, B, C, bias = torch.randn(M*K, device="cuda"), torch.randn(N*K, device="cuda"), torch.zeros(M*N, device="cuda"), torch.randn(N, device="cuda")
g = gemm(M, N, K, L.ptr(A), K, 1, L.ptr(B), K, 1, L.ptr(C), N, 2, L.ptr(bias))
for _ in range(5):
    L.call("addhip_gemm_f32", g, L.current_stream())
torch.cuda.synchronize()

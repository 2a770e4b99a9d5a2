"""addhip_actor_head alone (csrc/actor_head.hip) at several row counts: per-row-block time vs fixed cost.  usage: actor_head_bench.py [hidden]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, add_gym_amd
import add_gym_amd._lib as L
K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = "cuda"
st = torch.cuda.current_stream()
for M in (32 * 256, 32 * 512, 32 * 1024, 32 * 2048):
    H = torch.relu(torch.randn(M, K, device=dev))
    Wh = torch.zeros(32, K, device=dev); Wh[:29] = torch.randn(29, K, device=dev) * 0.05
    bh = torch.zeros(32, device=dev)
    na = torch.zeros(M, 32, device=dev); na[:, :29] = torch.randn(M, 29, device=dev)
    ol, adv, mask, nv = torch.randn(M, device=dev) * 0.3 - 30, torch.randn(M, device=dev), torch.ones(M, device=dev), torch.full((1,), float(M), device=dev)
    ns = L.load().addhip_actor_head_slabs(M)
    dz, slabs, gb, stats = torch.zeros(M, K, device=dev), torch.zeros(ns, L.actor_head_slab(K), device=dev), torch.zeros(16, K, device=dev), torch.zeros(8, device=dev)
    h = L.ActorHeadT(M, K, L.ptr(H), L.ptr(Wh), L.ptr(bh), L.ptr(na), L.ptr(ol), L.ptr(adv), L.ptr(mask), L.ptr(nv), 0.05, 40.0, 0.2, 10.0, 0.0, 1.0, None,
                     L.ptr(dz), None, 0, L.ptr(slabs), ns, L.ptr(gb), 16, K, L.ptr(stats), None)
    for _ in range(5): L.call("addhip_actor_head", h, st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): L.call("addhip_actor_head", h, st.cuda_stream)
    e1.record(st); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"hidden={K} rows={M} ({M // 32 // ns} row blocks per workgroup, {ns} workgroups): {us:7.1f} us", flush=True)

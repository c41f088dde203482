"""GEMM micro-benchmark on the shapes of the LLaVA-1.5-7B step (random data, HIP events, interleaved variants).
Usage (GPU box): python tools/gemm_bench.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8704
dev = torch.device("cuda:0")
d, I, V = 4096, 11008, 32064
# (name, layout, M, N, K)  -- logical C[M,N], reduction K
shapes = [
    ("qkv fwd", hip.NT, M, 3 * d, d), ("o fwd", hip.NT, M, d, d), ("gate_up fwd", hip.NT, M, 2 * I, d),
    ("down fwd", hip.NT, M, d, I), ("lm_head fwd", hip.NT, M, V, d),
    ("qkv dgrad", hip.NN, M, d, 3 * d), ("o dgrad", hip.NN, M, d, d), ("gate_up dgrad", hip.NN, M, d, 2 * I),
    ("down dgrad", hip.NN, M, I, d), ("lm_head dgrad", hip.NN, M, d, V),
    ("qkv wgrad", hip.TN, 3 * d, d, M), ("o wgrad", hip.TN, d, d, M), ("gate_up wgrad", hip.TN, 2 * I, d, M),
    ("down wgrad", hip.TN, d, I, M), ("lm_head wgrad", hip.TN, V, d, M),
]
g = torch.Generator(device=dev).manual_seed(0)
tot = {128: 0.0, 256: 0.0, 0: 0.0}
flops_tot = 0.0
for name, layout, m, n, k in shapes:
    if layout == hip.NT:
        A = torch.randn(m, k, device=dev, generator=g).bfloat16(); B = torch.randn(n, k, device=dev, generator=g).bfloat16()
    elif layout == hip.NN:
        A = torch.randn(m, k, device=dev, generator=g).bfloat16(); B = torch.randn(k, n, device=dev, generator=g).bfloat16()
    else:
        A = torch.randn(k, m, device=dev, generator=g).bfloat16(); B = torch.randn(k, n, device=dev, generator=g).bfloat16()
    C = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    res = {}
    for rnd in range(3):
        for tile in (128, 256, 0):
            hip.gemm_set_tile(tile)
            hip.gemm(layout, A, B, C, accumulate=(layout == hip.TN))
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                hip.gemm(layout, A, B, C, accumulate=(layout == hip.TN))
            e1.record(); torch.cuda.synchronize()
            res.setdefault(tile, []).append(e0.elapsed_time(e1) / 5)
    fl = 2.0 * m * n * k
    flops_tot += fl
    line = f"{name:16s} {m:6d}x{n:6d}x{k:6d}"
    for tile in (128, 256, 0):
        ms = min(res[tile]); tot[tile] += ms
        line += f" | t{tile}: {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF"
    print(line, flush=True)
for tile in (128, 256, 0):
    print(f"sum tile {tile}: {tot[tile]:.3f} ms -> {flops_tot/tot[tile]/1e9:.1f} TF aggregate")

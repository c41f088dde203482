"""Sweep GROUP_M (tile rasterisation) for a few GEMM cases."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8704
cases = [("gate_up wgrad", hip.TN, 22016, 4096, M), ("down wgrad", hip.TN, 4096, 11008, M), ("qkv wgrad", hip.TN, 12288, 4096, M),
         ("lm_head wgrad", hip.TN, 32064, 4096, M), ("gate_up fwd", hip.NT, M, 22016, 4096), ("gate_up dgradT", hip.NT, M, 4096, 22016)]
g = torch.Generator(device=dev).manual_seed(0)
hip.gemm_set_tile(256)
for name, layout, m, n, k in cases:
    if layout == hip.NT:
        A = torch.randn(m, k, device=dev, generator=g).bfloat16(); B = torch.randn(n, k, device=dev, generator=g).bfloat16()
    else:
        A = torch.randn(k, m, device=dev, generator=g).bfloat16(); B = torch.randn(k, n, device=dev, generator=g).bfloat16()
    C = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    line = f"{name:16s} {m}x{n}x{k}"
    for gm in (1, 2, 4, 8, 16, 32):
        hip.gemm_set_group_m(gm)
        best = 1e9
        for rnd in range(3):
            hip.gemm(layout, A, B, C)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                hip.gemm(layout, A, B, C)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 4)
        line += f" | g{gm}: {2.0*m*n*k/best/1e9:6.0f}"
    print(line, flush=True)

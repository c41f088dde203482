"""The CLIP tower's GEMMs (K = 1024 / 4096, M = images x 577) and the projector's under the 256^2 and the 128^2 kernel,
with the epilogues the step uses; interleaved rounds in one process.  Usage (GPU box): python tools/clip_gemm_tiles.py [images]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
hip.ensure_gemm_workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
M = n_img * 577
MP = n_img * 576
cases = [("qkv  bias", M, 3072, 1024, "b"), ("out  bias+resid", M, 1024, 1024, "br"), ("fc1  bias+quick_gelu", M, 4096, 1024, "bg"),
         ("fc2  bias+resid", M, 1024, 4096, "br"), ("patch embed", MP, 1024, 608, ""), ("proj1 bias+gelu", MP, 4096, 1024, "bG"),
         ("proj2 bias", MP, 4096, 4096, "b")]
for name, m, n, k, epi in cases:
    A = torch.randn(m, k, device=dev, generator=g).bfloat16()
    B = (torch.randn(n, k, device=dev, generator=g) * 0.02).bfloat16()
    C = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(n, device=dev, generator=g).bfloat16() if "b" in epi else None
    resid = torch.randn(m, n, device=dev, generator=g).bfloat16() if "r" in epi else None
    act = hip.ACT_QUICK_GELU if "g" in epi else (hip.ACT_GELU if "G" in epi else 0)
    res = {128: [], 256: []}
    outs = {}
    for rnd in range(4):
        for tile in (256, 128):
            hip.gemm_set_tile(tile)
            hip.gemm(hip.NT, A, B, C, bias=bias, resid=resid, act=act)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                hip.gemm(hip.NT, A, B, C, bias=bias, resid=resid, act=act)
            e1.record(); torch.cuda.synchronize()
            res[tile].append(e0.elapsed_time(e1) / 10 * 1e3)
            outs[tile] = C.float().clone()
    hip.gemm_set_tile(0)
    pick = 256 if hip.load().mmu_gemm_uses_256(m, n, k) else 128
    fl = 2.0 * m * n * k
    t2, t1 = statistics.median(res[256]), statistics.median(res[128])
    print(f"{name:22s} {m:6d}x{n:5d}x{k:5d}  256^2 {t2:7.1f} us {fl/t2/1e6:7.1f} TF | 128^2 {t1:7.1f} us {fl/t1/1e6:7.1f} TF | heuristic picks {pick}"
          f" | max diff {(outs[256]-outs[128]).abs().max().item():.3g}", flush=True)

"""XCD start skew of the 256^2 GEMM (mmu_gemm_set_xcd_skew): the step's heaviest shapes at M = 32640, skew 0..N units
interleaved in one process (HIP events, random data), incl. the GEMMs with the SwiGLU epilogues (heaviest epilogue traffic).
Usage (GPU box): python tools/gemm_xcd_skew.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32640
dev = torch.device("cuda:0")
d, I = 4096, 11008
hip.ensure_gemm_workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).bfloat16()
SKEWS = (0, 1, 2, 3, 4, 6, 8)
cases = []
x, wqkv, wo, wgu, wdT = rnd(M, d), rnd(3 * d, d), rnd(d, d), rnd(2 * I, d), rnd(I, d)
qkv, o, gu, act, actT = (torch.empty(M, 3 * d, device=dev, dtype=torch.bfloat16), torch.empty(M, d, device=dev, dtype=torch.bfloat16),
                         torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16), torch.empty(M, I, device=dev, dtype=torch.bfloat16),
                         torch.empty(I, M, device=dev, dtype=torch.bfloat16))
dgu, dguT = torch.empty(M, 2 * I, device=dev, dtype=torch.bfloat16), torch.empty(2 * I, M, device=dev, dtype=torch.bfloat16)
resid = rnd(M, d)
cases.append(("qkv fwd (store)", 2.0 * M * 3 * d * d, lambda: hip.gemm(hip.NT, x, wqkv, qkv)))
cases.append(("o fwd (residual)", 2.0 * M * d * d, lambda: hip.gemm(hip.NT, x, wo, o, resid=resid)))
cases.append(("gate_up fwd (store)", 2.0 * M * 2 * I * d, lambda: hip.gemm(hip.NT, x, wgu, gu)))
cases.append(("gate_up fwd + swiglu", 2.0 * M * 2 * I * d, lambda: hip.gemm_swiglu_fwd(x, wgu, gu, act, actT)))
cases.append(("down dgrad + swiglu bwd", 2.0 * M * I * d, lambda: hip.gemm_swiglu_bwd(x, wdT, True, gu, dgu, dguT)))
xT, dyT, gacc = rnd(d, M), rnd(3 * d, M), torch.zeros(3 * d, d, device=dev, dtype=torch.bfloat16)
cases.append(("qkv wgrad (accumulate)", 2.0 * M * 3 * d * d, lambda: hip.gemm(hip.NT, dyT, xT, gacc, accumulate=True)))
for name, fl, fn in cases:
    best = {s: 1e9 for s in SKEWS}
    for rep in range(3):
        for s in SKEWS:
            hip.gemm_set_xcd_skew(s)
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best[s] = min(best[s], e0.elapsed_time(e1) / 5)
    hip.gemm_set_xcd_skew(0)
    print(f"{name:26s} " + "  ".join(f"skew {s}: {best[s]*1e3:7.1f} us ({fl/best[s]/1e9:6.1f} TF)" for s in SKEWS), flush=True)

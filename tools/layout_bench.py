"""Layout / fused-transpose kernels at the 7B step's shapes (M = 15 x 1088 tokens): time and effective HBM bandwidth
(algorithmic bytes = every input read once + every output written once).  Usage: [MMU_HIP_LIB=...] python tools/layout_bench.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip
dev = torch.device("cuda:0")
M, d, I = 15 * 1088, 4096, 11008
BF = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
gu = torch.randn(M, 2 * I, device=dev, generator=g).to(BF)
dh = torch.randn(M, I, device=dev, generator=g).to(BF)
x = torch.randn(M, d, device=dev, generator=g).to(BF)
w = torch.randn(d, device=dev, generator=g).to(BF)
rstd = torch.rand(M, device=dev, generator=g) + 0.5
dgu, dguT = torch.empty(M, 2 * I, device=dev, dtype=BF), torch.empty(2 * I, M, device=dev, dtype=BF)
act, actT = torch.empty(M, I, device=dev, dtype=BF), torch.empty(I, M, device=dev, dtype=BF)
yT = torch.empty(d, M, device=dev, dtype=BF)
xT = torch.empty(d, M, device=dev, dtype=BF)
guT = torch.empty(2 * I, M, device=dev, dtype=BF)
cp = torch.empty_like(gu)
B = 2
cases = [
    ("swiglu_bwd_t   (fused_t<2>)", lambda: hip.swiglu_bwd_t(gu, dh, dgu, dguT), (2 * I + I + 2 * I + 2 * I) * M * B),
    ("swiglu_fwd_both (fused_t<3>)", lambda: hip.swiglu_fwd_both(gu, act, actT), (2 * I + I + I) * M * B),
    ("swiglu_fwd_t   (fused_t<0>)", lambda: hip.swiglu_fwd_t(gu, actT), (2 * I + I) * M * B),
    ("rmsnorm_apply_t (fused_t<1>)", lambda: hip.rmsnorm_apply_t(x, w, rstd, yT), (d + d) * M * B),
    ("transpose [M, 4096]", lambda: hip.transpose(x, xT), 2 * d * M * B),
    ("transpose [M, 22016]", lambda: hip.transpose(gu, guT), 2 * 2 * I * M * B),
    ("torch copy [M, 22016] (reference stream)", lambda: cp.copy_(gu), 2 * 2 * I * M * B),
]
HOT = "--hot" in sys.argv  # a burst of GEMMs in front of every timed launch: the power / clock state of the step
if HOT:
    ga = torch.randn(M, d, device=dev, generator=g).to(BF); gb = torch.randn(I, d, device=dev, generator=g).to(BF)
    gc = torch.empty(M, I, device=dev, dtype=BF)
def timed_hot(fn, n=5):
    tot = 0.0
    for _ in range(n):
        for _ in range(6): hip.gemm(hip.NT, ga, gb, gc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3
def timed(fn, n=10):
    if HOT: return timed_hot(fn)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("lib:", os.environ.get("MMU_HIP_LIB", "default"))
for name, fn, nbytes in cases:
    fn(); fn()
    ts = [timed(fn) for _ in range(5)]
    t = statistics.median(ts)
    print(f"{name:44s} {t:8.1f} us  {nbytes / t / 1e6:6.2f} TB/s")

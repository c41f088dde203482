"""A/B of the 256x256 kernel's launch modes (0 workgroup per tile, 1 persistent) on the shapes of the step; checks the
persistent result against mode 0 bit for bit.  (Round-1 finding with a debug build that skipped the epilogue stores:
the stores cost 7-9 % of a K=4096 GEMM, i.e. the C write-back at ~5 TB/s is serialised with the compute.)  Usage (GPU box): python tools/gemm_persist.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip

dev = torch.device("cuda:0")
M, d, I = 32640, 4096, 11008
shapes = [("qkv fwd", M, 3 * d, d, False), ("o fwd", M, d, d, False), ("gate_up fwd", M, 2 * I, d, False),
          ("down fwd", M, d, I, False), ("down dgrad", M, I, d, False), ("gate_up wgrad", 2 * I, d, M, True),
          ("down wgrad", d, I, M, True), ("ragged", 1000, 1096, 520, False)]
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, reps=5):
    fn()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


for name, m, n, k, acc in shapes:
    A = torch.randn(m, k, device=dev, generator=g).bfloat16()
    B = torch.randn(n, k, device=dev, generator=g).bfloat16()
    C0 = torch.randn(m, n, device=dev, generator=g).bfloat16()
    hip.gemm_set_tile(256)
    outs, line = {}, f"{name:14s} {m:6d}x{n:6d}x{k:6d}"
    for mode in (0, 1):
        hip.gemm_set_persistent(mode)
        C = C0.clone()
        hip.gemm(hip.NT, A, B, C, accumulate=acc)
        outs[mode] = C.clone()
        t = timeit(lambda: hip.gemm(hip.NT, A, B, C, accumulate=acc))
        line += f" | m{mode}: {t*1e3:8.1f} us {2.0*m*n*k/t/1e9:7.1f} TF"
    ok = torch.equal(outs[0], outs[1])
    print(line, "| equal" if ok else "| MISMATCH", flush=True)
    hip.gemm_set_persistent(0)
    del A, B, C0, C

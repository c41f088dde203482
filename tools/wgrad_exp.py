"""Experiment: wgrad as TN (token-major operands) vs transposes + NT, and TN sensitivity to operand size / pitch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
hip.gemm_set_tile(256)

def timeit(fn, n=4):
    best = 1e9
    for _ in range(3):
        fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best

for (Nw, Kw, M) in [(22016, 4096, 32640), (4096, 11008, 32640), (4096, 4096, 32640), (12288, 4096, 32640),
                    (22016, 4096, 2176), (4096, 4096, 8704)]:
    dy = torch.randn(M, Nw, device=dev, generator=g).bfloat16()
    x = torch.randn(M, Kw, device=dev, generator=g).bfloat16()
    dw = torch.zeros(Nw, Kw, device=dev, dtype=torch.bfloat16)
    dyT = torch.empty(Nw, M, device=dev, dtype=torch.bfloat16)
    xT = torch.empty(Kw, M, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * Nw * Kw
    t_tn = timeit(lambda: hip.gemm(hip.TN, dy, x, dw, accumulate=True))
    t_tr = timeit(lambda: (hip.transpose(dy, dyT), hip.transpose(x, xT)))
    t_nt = timeit(lambda: hip.gemm(hip.NT, dyT, xT, dw, accumulate=True))
    gb = (dy.numel() + x.numel()) * 2 * 2 / 1e9
    print(f"wgrad {Nw}x{Kw} M={M}: TN {t_tn*1e3:8.1f} us {fl/t_tn/1e9:6.0f} TF | transposes {t_tr*1e3:7.1f} us ({gb/t_tr*1e3:5.0f} GB/s) | "
          f"NT {t_nt*1e3:8.1f} us {fl/t_nt/1e9:6.0f} TF | NT+tr {fl/(t_nt+t_tr)/1e9:6.0f} TF", flush=True)

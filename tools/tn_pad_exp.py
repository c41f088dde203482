"""Why is the reduction-MAJOR (TN) operand path of the 256^2 GEMM ~35 % slower than NT on the wgrad shapes (DESIGN
'Layout policy')?  Hypothesis: activations have power-of-two row strides (8 KiB / 24 KiB), so the 256-B runs a workgroup
fetches from consecutive k-rows of a K-major tile fall on a handful of HBM channels at any instant (latency-bound queues),
while a K-contiguous operand is swept along its rows over all channels.  Test: the same TN GEMM with the operands' leading
dimensions padded by 64 / 128 / 192 elements (views into wider buffers), against NT on pre-transposed copies.
Usage: python tools/tn_pad_exp.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip

dev = torch.device("cuda:0")
hip.ensure_gemm_workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
BF = torch.bfloat16


def t(fn, n=6):
    fn(); torch.cuda.synchronize()
    v = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        v.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(v)


for name, (Mtok, Nout, Kin) in (("qkv wgrad", (32640, 12288, 4096)), ("o wgrad", (32640, 4096, 4096)), ("down wgrad", (32640, 4096, 11008))):
    fl = 2.0 * Mtok * Nout * Kin
    line = [name]
    for pad in (0, 64, 128, 192, 320):
        dy_buf = torch.randn(Mtok, Nout + pad, device=dev, generator=g).to(BF)
        x_buf = torch.randn(Mtok, Kin + pad, device=dev, generator=g).to(BF)
        dy, x = dy_buf[:, :Nout], x_buf[:, :Kin]
        out = torch.zeros(Nout, Kin, device=dev, dtype=BF)
        us = t(lambda: hip.gemm(hip.TN, dy, x, out, accumulate=True))
        line.append(f"TN pad{pad}: {us:.0f} us {fl/us/1e6:.0f} TF")
        del dy_buf, x_buf
    dyT = torch.randn(Nout, Mtok, device=dev, generator=g).to(BF)
    xT = torch.randn(Kin, Mtok, device=dev, generator=g).to(BF)
    out = torch.zeros(Nout, Kin, device=dev, dtype=BF)
    us = t(lambda: hip.gemm(hip.NT, dyT, xT, out, accumulate=True))
    line.append(f"NT (pre-transposed): {us:.0f} us {fl/us/1e6:.0f} TF")
    print(" | ".join(line), flush=True)

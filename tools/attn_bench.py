"""Attention kernel micro-benchmark at the 7B step's shape (B sequences x 32 heads x S=1088 x D=128, causal)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H, S, D = 32, 1088, 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
HD = H * D
qkv = torch.randn(B * S, 3 * HD, device=dev, generator=g).bfloat16()
o = torch.empty(B * S, HD, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
do = torch.randn(B * S, HD, device=dev, generator=g).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty(2, B, H, S, device=dev, dtype=torch.float32)
sc = 1 / math.sqrt(D)
def fwd(): hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2*HD], qkv[:, 2*HD:], o, lse, None, B, S, H, D, sc, True)
def bwd(): hip.attn_bwd(qkv[:, :HD], qkv[:, HD:2*HD], qkv[:, 2*HD:], o, lse, None, do, delta, dqkv[:, :HD], dqkv[:, HD:2*HD], dqkv[:, 2*HD:], B, S, H, D, sc, True)
def t(fn, n=5):
    best = 1e9
    for _ in range(3):
        fn(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / n)
    return best
fl = 2 * 2 * B * H * S * S * D / 2
tf, tb = t(fwd), t(bwd)
print(f"B={B}: fwd {tf*1e3:.0f} us {fl/tf/1e9:.0f} TF | bwd(delta+dkdv+dq) {tb*1e3:.0f} us {2.5*fl/tb/1e9:.0f} TF (5-product algorithmic)")

"""Run ONE GEMM case repeatedly (for rocprofv3 --pmc passes). python tools/gemm_case.py LAYOUT M N K [tile] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip
layout = {"NT": 0, "NN": 1, "TN": 2}[sys.argv[1]]
m, n, k = (int(x) for x in sys.argv[2:5])
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 256
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
if layout == 0:
    A = torch.randn(m, k, device=dev, generator=g).bfloat16(); B = torch.randn(n, k, device=dev, generator=g).bfloat16()
elif layout == 1:
    A = torch.randn(m, k, device=dev, generator=g).bfloat16(); B = torch.randn(k, n, device=dev, generator=g).bfloat16()
else:
    A = torch.randn(k, m, device=dev, generator=g).bfloat16(); B = torch.randn(k, n, device=dev, generator=g).bfloat16()
C = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
hip.gemm_set_tile(tile)
for _ in range(iters):
    hip.gemm(layout, A, B, C)
torch.cuda.synchronize()
print("done", sys.argv[1:])

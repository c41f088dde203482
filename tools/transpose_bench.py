import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip
dev = torch.device("cuda:0")
for (R, C) in ((32640, 4096), (32640, 12288), (4096, 11008)):
    x = torch.randn(R, C, device=dev).bfloat16(); y = torch.empty(C, R, device=dev, dtype=torch.bfloat16)
    hip.transpose(x, y); assert torch.equal(y, x.t().contiguous())
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): hip.transpose(x, y)
        e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) / 10)
    print(R, C, f"{best*1e3:.1f} us {2*R*C*2/best/1e9:.2f} TB/s")

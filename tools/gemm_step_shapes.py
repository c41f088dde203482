"""The GEMMs of one 7B micro-batch exactly as the step issues them (all NT: forward, dgrad through the resident W^T,
wgrad through transposed activations with the accumulate epilogue), at M = 32640 tokens.  Prints TF per shape, the tile
grid and its rounds over 256 CUs, so that tail / prologue losses are visible per shape.
Usage (GPU box): python tools/gemm_step_shapes.py [M]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32640
dev = torch.device("cuda:0")
d, I, V = 4096, 11008, 32064
shapes = []
for nm, n_out, n_in in (("qkv", 3 * d, d), ("o", d, d), ("gate_up", 2 * I, d), ("down", d, I)):
    shapes.append((nm + " fwd", M, n_out, n_in, False, 32))
    shapes.append((nm + " dgrad", M, n_in, n_out, False, 32))
    shapes.append((nm + " wgrad", n_out, n_in, M, True, 32))
shapes.append(("lm_head fwd", M, V, d, False, 1))
shapes.append(("lm_head dgrad", M, d, V, False, 1))
shapes.append(("lm_head wgrad", V, d, M, True, 1))
hip.ensure_gemm_workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
tot_t = tot_f = 0.0
for name, m, n, k, acc, mult in shapes:
    A = torch.randn(m, k, device=dev, generator=g).bfloat16()
    B = torch.randn(n, k, device=dev, generator=g).bfloat16()
    C = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    best, best_off = 1e9, 1e9
    for rnd in range(3):
        for sk in (0, 1):
            hip.gemm_set_tail_split(sk)
            hip.gemm(hip.NT, A, B, C, accumulate=acc)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                hip.gemm(hip.NT, A, B, C, accumulate=acc)
            e1.record(); torch.cuda.synchronize()
            if sk:
                best = min(best, e0.elapsed_time(e1) / 5)
            else:
                best_off = min(best_off, e0.elapsed_time(e1) / 5)
    fl = 2.0 * m * n * k
    tiles = math.ceil(m / 256) * math.ceil(n / 256)
    rounds = tiles / 256
    eff_rounds = math.ceil(rounds)
    tot_t += best * mult; tot_f += fl * mult
    print(f"{name:15s} {m:6d}x{n:6d}x{k:6d} {best*1e3:8.1f} us {fl/best/1e9:7.1f} TF  tiles {tiles:5d} = {rounds:6.2f} rounds"
          f" (quantisation {rounds/eff_rounds:.3f})  tail split off {best_off*1e3:8.1f} us  k-iters {k//64}", flush=True)
    del A, B, C
print(f"weighted micro-batch GEMM time {tot_t:.1f} ms -> {tot_f/tot_t/1e9:.1f} TF aggregate")

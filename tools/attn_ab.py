"""A/B of two builds of libmmu_hip.so on the attention kernels at the 7B step's shape (B x 32 heads x S 1088 x D 128,
causal), interleaved rounds in ONE process (guide rule 24), plus a cross-check of the outputs.
Usage: python tools/attn_ab.py LIB_A LIB_B [B] [mask]"""
import ctypes as C, math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip

# LIB or LIB:VARIANT (dK/dV kernel variant 8 / 4 of that library)
la, lb = sys.argv[1], sys.argv[2]
var = {}
for tag, spec in (("A", la), ("B", lb)):
    if ":" in spec:
        spec, v = spec.rsplit(":", 1)
        var[tag] = int(v)
    if tag == "A": la = spec
    else: lb = spec
B = int(sys.argv[3]) if len(sys.argv) > 3 else 30
ONES_MASK = len(sys.argv) > 4 and sys.argv[4] == "mask"  # a device-resident all-ones key mask, as the step passes
H, S, D = 32, 1088, 128
dev = torch.device("cuda:0")
libs = {}
for tag, path in (("A", la), ("B", lb)):
    lib = C.CDLL(path)
    for n in ("mmu_attn_fwd", "mmu_attn_bwd_rope"):
        f = getattr(lib, n); f.argtypes = hip.SIGNATURES[n]; f.restype = C.c_int
    libs[tag] = lib
    if tag in var:
        lib.mmu_attn_set_dkdv_variant.argtypes = [C.c_int]; lib.mmu_attn_set_dkdv_variant.restype = C.c_int
g = torch.Generator(device=dev).manual_seed(0)
HD = H * D
qkv = torch.randn(B * S, 3 * HD, device=dev, generator=g).bfloat16()
do = torch.randn(B * S, HD, device=dev, generator=g).bfloat16()
sc = 1 / math.sqrt(D)
kmask = torch.ones(B, S, device=dev, dtype=torch.int32)
KM = kmask.data_ptr() if ONES_MASK else 0
st = lambda: torch.cuda.current_stream().cuda_stream
out = {}
for tag in libs:
    out[tag] = dict(o=torch.empty(B * S, HD, device=dev, dtype=torch.bfloat16), lse=torch.empty(B, H, S, device=dev),
                    dqkv=torch.empty_like(qkv), delta=torch.empty(2, B, H, S, device=dev))
p = lambda t: t.data_ptr()
def fwd(tag):
    o = out[tag]
    rc = libs[tag].mmu_attn_fwd(p(qkv), p(qkv) + 2 * HD, p(qkv) + 4 * HD, 3 * HD, p(o["o"]), HD, p(o["lse"]), KM, B, S, H, D, sc, 1, st())
    assert rc == 0, rc
def bwd(tag):
    o = out[tag]
    if tag in var:
        assert libs[tag].mmu_attn_set_dkdv_variant(var[tag]) == 0
    d = o["dqkv"]
    rc = libs[tag].mmu_attn_bwd_rope(p(qkv), p(qkv) + 2 * HD, p(qkv) + 4 * HD, 3 * HD, p(o["o"]), HD, p(o["lse"]), KM, p(do), HD,
                                     p(o["delta"]), p(d), p(d) + 2 * HD, p(d) + 4 * HD, 3 * HD, B, S, H, D, sc, 1, 0, 0, 0, st())
    assert rc == 0, rc
def timed(fn, tag, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn(tag)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for tag in libs: fwd(tag); bwd(tag)
res = {(k, t): [] for k in ("fwd", "bwd") for t in libs}
for r in range(8):
    for tag in libs:
        res[("fwd", tag)].append(timed(fwd, tag)); res[("bwd", tag)].append(timed(bwd, tag))
fl = 2 * 2 * B * H * S * S * D / 2
for k, mult in (("fwd", 1.0), ("bwd", 2.5)):
    for tag in libs:
        v = res[(k, tag)]
        print(f"{k} {tag} ({os.path.basename(la if tag=='A' else lb)}): median {statistics.median(v):.0f} us min {min(v):.0f} us = "
              f"{mult*fl/statistics.median(v)/1e6:.0f} TFLOP/s (algorithmic, causal-counted{', 5 products' if k=='bwd' else ''})")
rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
print("A vs B: o", rel(out["A"]["o"], out["B"]["o"]), "dqkv", rel(out["A"]["dqkv"], out["B"]["dqkv"]))

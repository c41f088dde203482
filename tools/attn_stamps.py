"""Where does a dK/dV workgroup spend its cycles?  Diagnostic build (-DMMU_ATTN_STAMP: s_memtime stamps around the
prologue, the tile sweep, the loop-top waits and the epilogue of attn_bwd_dkdv_kernel, written to a buffer of their own;
the real library has no stamp; the diagnostic build's run time is not quoted, only its shares).
Usage: python tools/attn_stamps.py [B]"""
import ctypes as C, os, statistics, subprocess, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "multimodal_unlearning_amd", "csrc", "variants", "libmmu_hip_attnstamp.so")
if not os.path.exists(lib):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "attnstamp", "-DMMU_ATTN_STAMP"], check=True)
os.environ["MMU_HIP_LIB"] = lib
import torch
from multimodal_unlearning_amd import hip
L = hip.load()
L.mmu_debug_read_attn_stamps.argtypes = [C.c_void_p, C.c_int]; L.mmu_debug_read_attn_stamps.restype = C.c_int
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
H, S, D = 32, 1088, 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
HD = H * D
qkv = torch.randn(B * S, 3 * HD, device=dev, generator=g).bfloat16()
o = torch.empty(B * S, HD, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=dev)
do = torch.randn(B * S, HD, device=dev, generator=g).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty(2, B, H, S, device=dev)
sc = 1 / math.sqrt(D)
hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2*HD], qkv[:, 2*HD:], o, lse, None, B, S, H, D, sc, True)
for _ in range(3):
    hip.attn_bwd(qkv[:, :HD], qkv[:, HD:2*HD], qkv[:, 2*HD:], o, lse, None, do, delta, dqkv[:, :HD], dqkv[:, HD:2*HD], dqkv[:, 2*HD:], B, S, H, D, sc, True)
torch.cuda.synchronize()
n = min(16384, 9 * H * B)
buf = (C.c_ulonglong * (4 * n))()
assert L.mmu_debug_read_attn_stamps(buf, n) == 0
rows = [(buf[4*i], buf[4*i+1], buf[4*i+2], buf[4*i+3] >> 8, buf[4*i+3] & 255) for i in range(n)]
rows = [r for r in rows if r[4] > 0]
by = {}
for pro, sweep, wait, total, it in rows:
    by.setdefault(it, []).append((pro, sweep - pro, wait, total - sweep, total))
print("iterations | workgroups | prologue | sweep (excl. prologue) | of which loop-top waits | epilogue | total   [shader cycles, medians]")
for it in sorted(by):
    v = by[it]; med = lambda k: statistics.median(x[k] for x in v)
    print(f"{it:10d} | {len(v):10d} | {med(0):8.0f} | {med(1):10.0f} ({med(1)/it:6.0f}/iter) | {med(2):8.0f} ({med(2)/max(1,it-1):6.0f}/iter) | {med(3):8.0f} | {med(4):8.0f}")
tot = sum(x[4] for v in by.values() for x in v)
print("shares of all workgroup cycles: prologue %.1f%%, sweep %.1f%% (loop-top waits %.1f%%), epilogue %.1f%%" % (
    100*sum(x[0] for v in by.values() for x in v)/tot, 100*sum(x[1] for v in by.values() for x in v)/tot,
    100*sum(x[2] for v in by.values() for x in v)/tot, 100*sum(x[3] for v in by.values() for x in v)/tot))

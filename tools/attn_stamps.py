"""Where does a dK/dV workgroup spend its cycles?  Diagnostic build (-DMMU_ATTN_STAMP: s_memtime stamps written to a
buffer of their own; the real library has no stamp; the diagnostic build's run time is not quoted, only its shares).
  python tools/attn_stamps.py [B]          eight-wave kernel: prologue, tile sweep, loop-top waits, epilogue per workgroup
  python tools/attn_stamps.py --dkdv4 [B]  persistent four-wave kernel: per step (body, waits), per item (epilogue parts)
  python tools/attn_stamps.py --dq [B]     dQ kernel: prologue, per block (loop-top wait, staging issue, compute), epilogue
  python tools/attn_stamps.py --fwd [B]    forward kernel: per 64-key block of a workgroup's last wave: loop-top wait, staging
                                           issue, QK^T until the row max, softmax + O rescale, P V issue
ATTN_STAMP_LIB=<path> selects another diagnostic build, e.g. one made with -DDKDV4_EXP=<bits> (timing experiments that
switch parts of the four-wave kernel off; see attention_dkdv4.hip)."""
import ctypes as C, os, statistics, subprocess, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.environ.get("ATTN_STAMP_LIB") or os.path.join(ROOT, "multimodal_unlearning_amd", "csrc", "variants", "libmmu_hip_attnstamp.so")
if not os.path.exists(lib):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "attnstamp", "-DMMU_ATTN_STAMP"], check=True)
os.environ["MMU_HIP_LIB"] = lib
import torch
from multimodal_unlearning_amd import hip
L = hip.load()
L.mmu_debug_read_attn_stamps.argtypes = [C.c_void_p, C.c_int]; L.mmu_debug_read_attn_stamps.restype = C.c_int
_pos = [x for x in sys.argv[1:] if not x.startswith("--")]
B = int(_pos[0]) if _pos else 30
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
if "--dq" in sys.argv:
    # dQ kernel, last wave of every workgroup (it overwrites the forward's stamp slots: the backward runs after it):
    # [blocks, loop-top waits, staging issue, block compute, prologue (entry -> loop), epilogue (loop end -> stores retired), total]
    L.mmu_debug_read_fwd_stamps.argtypes = [C.c_void_p, C.c_int]; L.mmu_debug_read_fwd_stamps.restype = C.c_int
    n = min(16384, 9 * H * B)
    buf = (C.c_ulonglong * (8 * n))()
    assert L.mmu_debug_read_fwd_stamps(buf, n) == 0
    rows = [tuple(buf[8*i:8*i+8]) for i in range(n)]
    rows = [r for r in rows if r[0] > 0 and r[7] == 1]
    by = {}
    for r in rows: by.setdefault(r[0], []).append(r)
    print(f"dQ kernel, {len(rows)} workgroups, wave 3 of each; medians in shader-clock ticks")
    print("blocks  wgs | workgroup total = prologue + blocks x (wait + stage + compute) + epilogue")
    for nb in sorted(by):
        g_ = by[nb]
        med = lambda f: statistics.median(f(x) for x in g_)
        print(f"{nb:5d} {len(g_):5d} | {med(lambda x: x[6]):8.0f} = {med(lambda x: x[4]):6.0f} + {nb} x ({med(lambda x: x[1]/x[0]):5.0f} + {med(lambda x: x[2]/x[0]):5.0f} + "
              f"{med(lambda x: x[3]/x[0]):6.0f}) + {med(lambda x: x[5]):6.0f}")
    tot = sum(x[6] for x in rows)
    sh = lambda k: 100 * sum(x[k] for x in rows) / tot
    print("shares of that wave's time: prologue %.1f%%, loop-top waits %.1f%%, staging issue %.1f%%, block compute %.1f%%, epilogue %.1f%%" % (sh(4), sh(1), sh(2), sh(3), sh(5)))
    sys.exit(0)
if "--fwd" in sys.argv:
    # forward kernel, last wave of every workgroup: [blocks, loop-top wait, stage issue, QK^T -> row max, softmax + rescale,
    # P V issue, total, masked blocks] in s_memtime ticks
    for _ in range(3):
        hip.attn_fwd(qkv[:, :HD], qkv[:, HD:2*HD], qkv[:, 2*HD:], o, lse, None, B, S, H, D, sc, True)
    torch.cuda.synchronize()
    L.mmu_debug_read_fwd_stamps.argtypes = [C.c_void_p, C.c_int]; L.mmu_debug_read_fwd_stamps.restype = C.c_int
    n = min(16384, 9 * H * B)
    buf = (C.c_ulonglong * (8 * n))()
    assert L.mmu_debug_read_fwd_stamps(buf, n) == 0
    rows = [tuple(buf[8*i:8*i+8]) for i in range(n)]
    rows = [r for r in rows if r[0] > 0]
    by = {}
    for r in rows: by.setdefault(r[0], []).append(r)
    print(f"forward kernel, {len(rows)} workgroups, wave 3 of each; per-block medians in shader-clock ticks")
    print("blocks  wgs | per block: total = wait + stage + QK^T(->max) + softmax/rescale + PV issue + rest | masked blocks | workgroup total")
    for nb in sorted(by):
        g_ = by[nb]
        med = lambda f: statistics.median(f(x) for x in g_)
        parts = [med(lambda x, k=k: x[k] / x[0]) for k in (1, 2, 3, 4, 5)]
        tot = med(lambda x: x[6] / x[0])
        print(f"{nb:5d} {len(g_):5d} | {tot:7.0f} = {parts[0]:6.0f} + {parts[1]:5.0f} + {parts[2]:6.0f} + {parts[3]:6.0f} + {parts[4]:6.0f} + {tot - sum(parts):6.0f} | "
              f"{med(lambda x: x[7]):3.0f} | {med(lambda x: x[6]):8.0f}")
    tot = sum(x[6] for x in rows)
    sh = lambda k: 100 * sum(x[k] for x in rows) / tot
    print("shares of that wave's time: loop-top wait %.1f%%, stage issue %.1f%%, QK^T %.1f%%, softmax + rescale %.1f%%, PV issue %.1f%%" % (sh(1), sh(2), sh(3), sh(4), sh(5)))
    sys.exit(0)
if "--dkdv4" in sys.argv:
    # the four-wave kernel's own stamps (wave 0 of each workgroup): 8 words per workgroup
    hip.attn_set_dkdv_variant(4)
    for _ in range(3):
        hip.attn_bwd(qkv[:, :HD], qkv[:, HD:2*HD], qkv[:, 2*HD:], o, lse, None, do, delta, dqkv[:, :HD], dqkv[:, HD:2*HD], dqkv[:, 2*HD:], B, S, H, D, sc, True)
    torch.cuda.synchronize()
    L.mmu_debug_read_dkdv4_stamps.argtypes = [C.c_void_p, C.c_int]; L.mmu_debug_read_dkdv4_stamps.restype = C.c_int
    # per persistent workgroup: [items, steps, vmcnt waits, barrier waits, total, item epilogues, clean step bodies, their count]
    n = 256
    buf = (C.c_ulonglong * (16 * n))()
    assert L.mmu_debug_read_dkdv4_stamps(buf, n) == 0
    rows = [tuple(buf[16*i:16*i+16]) for i in range(n)]
    rows = [r for r in rows if r[0] > 0]
    med = lambda f: statistics.median(f(x) for x in rows)
    print(f"workgroups {len(rows)}: items {med(lambda x: x[0]):.0f}, steps {med(lambda x: x[1]):.0f}, total {med(lambda x: x[4]):.0f} ticks "
          f"(min {min(x[4] for x in rows)}, max {max(x[4] for x in rows)})  [s_memtime ticks, wave 0, medians over workgroups]")
    print(f"per step: {med(lambda x: x[4]/x[1]):.0f} total | vmcnt wait {med(lambda x: x[2]/x[1]):.0f} | barrier wait {med(lambda x: x[3]/x[1]):.0f} | "
          f"clean step body {med(lambda x: x[6]/max(1,x[7])):.0f} ({med(lambda x: x[7]/x[1])*100:.0f}% of steps)")
    print(f"per item: epilogue {med(lambda x: x[5]/x[0]):.0f} = scale/RoPE/bf16 {med(lambda x: x[8]/x[0]):.0f} + next K/V request {med(lambda x: x[9]/x[0]):.0f} "
          f"+ swaps/stores {med(lambda x: x[10]/x[0]):.0f} + counted wait {med(lambda x: (x[5]-x[8]-x[9]-x[10])/x[0]):.0f}")
    print(f"masked / partial steps: {med(lambda x: x[11]/max(1,x[1]-x[7])):.0f} per step ({med(lambda x: (x[1]-x[7])/x[1])*100:.0f}% of steps)")
    tot = sum(x[4] for x in rows)
    sh = lambda k: 100 * sum(x[k] for x in rows) / tot
    print("shares of workgroup time: vmcnt waits %.1f%%, barrier waits %.1f%%, clean bodies %.1f%%, item epilogues %.1f%%" % (sh(2), sh(3), sh(6), sh(5)))
    sys.exit(0)
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

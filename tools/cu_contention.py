"""What a kernel that occupies some compute units — as RCCL's all-reduce kernels do while the gradient exchange overlaps
the last backward sweep — costs the persistent kernels of the step, on ONE GPU: `tests/c_abi/squatter.hip` parks G
workgroups of 256 threads for a bounded time on a side stream while the main stream runs the step's GEMM shape
(persistent launch with the tile queue / with a static walk / one workgroup per tile) and the attention backward (four-wave persistent dK/dV vs eight-wave).
A persistent workgroup of these kernels owns the whole register file of a CU, so it cannot share a CU with a squatter.
Usage: python tools/cu_contention.py [out.json]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_unlearning_amd import hip
_so = os.path.join(ROOT, "tests", "c_abi", "libsquatter.so")
if not os.path.exists(_so):
    import subprocess
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", _so,
                    os.path.join(ROOT, "tests", "c_abi", "squatter.hip")], check=True)
sq = C.CDLL(_so)
sq.squat_launch.argtypes = [C.c_int, C.c_int, C.c_longlong, C.c_void_p]
sq.squat_launch.restype = C.c_int
dev = torch.device("cuda:0")
hip.load()
hip.ensure_gemm_workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream()
M, N, K = 32640, 12288, 4096
a = torch.randn(M, K, device=dev, generator=g).bfloat16()
b = torch.randn(N, K, device=dev, generator=g).bfloat16()
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
B, S, H, D = 30, 1088, 32, 128
qkv = (torch.randn(B * S, 3 * H * D, device=dev, generator=g) * 0.5).bfloat16()
o = torch.empty(B * S, H * D, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B, H, S, device=dev, dtype=torch.float32)
do = (torch.randn(B * S, H * D, device=dev, generator=g) * 0.1).bfloat16()
dqkv = torch.empty_like(qkv)
delta = torch.empty(2, B, H, S, device=dev, dtype=torch.float32)
dt = H * D
hip.attn_fwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], o, lse, None, B, S, H, D, D ** -0.5, True)

def run_gemm():
    hip.gemm(hip.NT, a, b, c)
def run_bwd():
    hip.attn_bwd(qkv[:, :dt], qkv[:, dt:2 * dt], qkv[:, 2 * dt:], o, lse, None, do, delta, dqkv[:, :dt], dqkv[:, dt:2 * dt],
                 dqkv[:, 2 * dt:], B, S, H, D, D ** -0.5, True)

def timed(fn, n, squat_blocks):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if squat_blocks:
        # the squatter is resident before the first timed launch and outlives the last (bounded: 60 ms per launch)
        with torch.cuda.stream(side):
            for _ in range(4):
                assert sq.squat_launch(squat_blocks, 256, 6_000_000, C.c_void_p(side.cuda_stream)) == 0
        time.sleep(0.002)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

res = []
for G in (0, 8, 16, 32, 64):
    row = {"squatter_workgroups": G}
    for name, mode in (("gemm_persistent_queue_us", 1), ("gemm_persistent_static_us", 2), ("gemm_per_tile_us", 0)):
        hip.gemm_set_persistent(mode)
        row[name] = round(timed(run_gemm, 20, G), 1)
    hip.gemm_set_persistent(1)
    for name, v in (("attn_bwd_dkdv4_us", 4), ("attn_bwd_dkdv8_us", 8)):
        hip.attn_set_dkdv_variant(v)
        row[name] = round(timed(run_bwd, 20, G), 1)
    hip.attn_set_dkdv_variant(4)
    print(json.dumps(row), flush=True)
    res.append(row)
if len(sys.argv) > 1:
    json.dump({"gemm": [M, N, K], "attention": [B, S, H, D], "note": "squatter: G workgroups x 256 threads parked for the "
               "whole timed region on a second stream", "results": res}, open(sys.argv[1], "w"), indent=1)

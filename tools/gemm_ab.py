"""A/B of two builds of libmmu_hip.so on GEMM shapes of the step (store / residual / accumulate epilogues), interleaved
in ONE process, random bf16 data.  Usage: python tools/gemm_ab.py LIB_A LIB_B"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_unlearning_amd import hip
dev = torch.device("cuda:0")
libs = {}
for tag, path in (("A", sys.argv[1]), ("B", sys.argv[2])):
    lib = C.CDLL(path)
    for n in ("mmu_gemm_bf16", "mmu_gemm_set_workspace", "mmu_gemm_set_tile_queue"):
        f = getattr(lib, n); f.argtypes = hip.SIGNATURES[n]; f.restype = C.c_int
    lib.mmu_gemm_workspace_bytes.restype = C.c_longlong
    lib.mmu_gemm_tile_queue_bytes.restype = C.c_longlong
    n, nq = lib.mmu_gemm_workspace_bytes(), lib.mmu_gemm_tile_queue_bytes()
    ws = torch.zeros(n + nq, device=dev, dtype=torch.uint8)
    assert lib.mmu_gemm_set_workspace(ws.data_ptr(), n) == 0 and lib.mmu_gemm_set_tile_queue(ws.data_ptr() + n, nq) == 0
    libs[tag] = (lib, ws)
g = torch.Generator(device=dev).manual_seed(0)
st = lambda: torch.cuda.current_stream().cuda_stream
for name, (M, N, K), epi in (("o fwd + residual", (32640, 4096, 4096), "resid"), ("down fwd + residual", (32640, 4096, 11008), "resid"),
                             ("qkv wgrad accumulate", (12288, 4096, 32640), "acc"), ("o fwd store", (32640, 4096, 4096), "store")):
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    b = torch.randn(N, K, device=dev, generator=g).bfloat16()
    r = torch.randn(M, N, device=dev, generator=g).bfloat16()
    out = {t: torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for t in libs}
    def run(t):
        lib = libs[t][0]
        rc = lib.mmu_gemm_bf16(0, a.data_ptr(), b.data_ptr(), out[t].data_ptr(), 0, r.data_ptr() if epi == "resid" else 0, M, N, K, K, K, N,
                               N if epi == "resid" else 0, 0, 1 if epi == "acc" else 0, st())
        assert rc == 0, rc
    res = {t: [] for t in libs}
    for t in libs: run(t)
    for _ in range(6):
        for t in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(t)
            e1.record(); torch.cuda.synchronize()
            res[t].append(e0.elapsed_time(e1) * 100)
    if epi != "acc":
        assert torch.equal(out["A"], out["B"])
    print(name, {t: round(statistics.median(v), 1) for t, v in res.items()}, "us", flush=True)
    del a, b, r, out

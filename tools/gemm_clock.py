"""Shader clock the chip HOLDS under the 256^2 GEMM (MI355X_MICROARCH.md 'DVFS give-back' item 6): a diagnostic build of
libmmu_hip.so (-DMMU_CLOCK_STAMP: s_memtime / s_memrealtime stamped once around each workgroup's tile loop, written to
a buffer nothing else reads) runs >= 2 s of back-to-back launches of one step shape on RANDOM data, then the last
launch's stamps are read: clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups.  The real library
has no stamp.  Usage: python tools/gemm_clock.py [out.json]   (builds the variant in-tree if missing)"""
import ctypes as C, json, os, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# GEMM_CLOCK_VARIANT / GEMM_CLOCK_FLAGS: the same measurement on another diagnostic build (tools/gemm_ceiling.py)
variant = os.environ.get("GEMM_CLOCK_VARIANT", "clock")
lib = os.path.join(ROOT, "multimodal_unlearning_amd", "csrc", "variants", f"libmmu_hip_{variant}.so")
if not os.path.exists(lib):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), variant, "-DMMU_CLOCK_STAMP",
                    *os.environ.get("GEMM_CLOCK_FLAGS", "").split()], check=True)
os.environ["MMU_HIP_LIB"] = lib
import torch
from multimodal_unlearning_amd import hip
L = hip.load()
L.mmu_debug_read_clock_stamps.argtypes = [C.c_void_p, C.c_int]
L.mmu_debug_read_clock_stamps.restype = C.c_int
dev = torch.device("cuda:0")
hip.ensure_gemm_workspace(dev)
g = torch.Generator(device=dev).manual_seed(0)
res = []
for name, (M, N, K) in (("qkv fwd 32640x12288x4096", (32640, 12288, 4096)), ("down fwd 32640x4096x11008", (32640, 4096, 11008)),
                        ("gate_up wgrad 22016x4096x32640", (22016, 4096, 32640))):
    a = torch.randn(M, K, device=dev, generator=g).bfloat16()
    b = torch.randn(N, K, device=dev, generator=g).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    hip.gemm(hip.NT, a, b, c)
    torch.cuda.synchronize()
    t0, n = time.time(), 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 2.5:
        for _ in range(20):
            hip.gemm(hip.NT, a, b, c)
        n += 20
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    buf = (C.c_ulonglong * 512)()
    assert L.mmu_debug_read_clock_stamps(buf, 256) == 0
    clk = [buf[2 * i] / buf[2 * i + 1] * 100.0 for i in range(256) if buf[2 * i + 1] > 0]
    r = {"shape": name, "launches": n, "avg_us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1),
         "clock_mhz_median": round(statistics.median(clk), 1), "clock_mhz_min": round(min(clk), 1), "clock_mhz_max": round(max(clk), 1),
         "mfma_peak_at_that_clock_tflops": round(256 * 4 * 1024 * statistics.median(clk) * 1e6 / 1e12, 1)}
    r["frac_of_peak_at_held_clock"] = round(r["tflops"] / r["mfma_peak_at_that_clock_tflops"], 3)
    print(json.dumps(r), flush=True)
    res.append(r)
    del a, b, c
if len(sys.argv) > 1:
    json.dump({"method": "s_memtime / s_memrealtime stamps around the tile loop of g256::gemm256_kernel, diagnostic build "
                         "(-DMMU_CLOCK_STAMP), >= 2.5 s of back-to-back launches on random bf16 data, last launch, median over 256 workgroups",
               "results": res}, open(sys.argv[1], "w"), indent=1)

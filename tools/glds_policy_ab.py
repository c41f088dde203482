"""Cache-policy bits of the LDS-DMA loads of the GEMM (sc0 / nt / sc1) against the default policy: diagnostic builds with
clock stamps, tools/gemm_clock.py timing, two interleaved passes.  Usage: python tools/glds_policy_ab.py OUTDIR
(--build-only on the CPU box first)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"clock": "", "glds_sc0": "-DMMU_GLDS_AUX=1", "glds_nt": "-DMMU_GLDS_AUX=2", "glds_sc1": "-DMMU_GLDS_AUX=16",
            "glds_sc0sc1": "-DMMU_GLDS_AUX=17"}
if sys.argv[1] == "--build-only":
    for name, flags in VARIANTS.items():
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), name, "-DMMU_CLOCK_STAMP", *flags.split()], check=True)
    sys.exit(0)
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
for rep in range(2):
    for name, flags in VARIANTS.items():
        env = dict(os.environ, GEMM_CLOCK_VARIANT=name, GEMM_CLOCK_FLAGS=flags)
        f = os.path.join(out, f"{name}_{rep}.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_clock.py"), f], env=env, check=True, stdout=subprocess.DEVNULL)
        for r in json.load(open(f))["results"]:
            print(name, rep, r["shape"], r["tflops"], "TF", r["clock_mhz_median"], "MHz", r["frac_of_peak_at_held_clock"], flush=True)

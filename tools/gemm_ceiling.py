"""Ceiling experiments on the 256x256 GEMM loop (diagnostic builds; their RESULTS ARE WRONG on purpose): what the loop
delivers, and at which held clock, when one of its costs is removed -
  noldsread: fragments are read from LDS in the first iteration only (MFMAs keep running on stale registers)
  nodma:     LDS-DMA staging in the first iteration only (no global / L2 traffic in the steady state)
  sametile:  every workgroup streams the panels of tile 0 (all staging served by L2)
next to the unmodified loop, all in one job on one box (tools/gemm_clock.py does the timing and the clock stamps).
Usage: python tools/gemm_ceiling.py OUTDIR     (build the variants first on the CPU box: --build-only)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"clock": "", "exp_noldsread": "-DMMU_EXP_NOLDSREAD", "exp_nodma": "-DMMU_EXP_NODMA", "exp_sametile": "-DMMU_EXP_SAMETILE",
            "exp_nodma_noldsread": "-DMMU_EXP_NODMA -DMMU_EXP_NOLDSREAD"}
if sys.argv[1] == "--build-only":
    for name, flags in VARIANTS.items():
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), name, "-DMMU_CLOCK_STAMP", *flags.split()], check=True)
    sys.exit(0)
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
for rep in range(2):  # two interleaved passes: the box's clock drifts with temperature
    for name, flags in VARIANTS.items():
        env = dict(os.environ, GEMM_CLOCK_VARIANT=name, GEMM_CLOCK_FLAGS=flags)
        f = os.path.join(out, f"{name}_{rep}.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_clock.py"), f], env=env, check=True)
        for r in json.load(open(f))["results"]:
            print(name, rep, r["shape"], r["tflops"], "TF", r["clock_mhz_median"], "MHz", r["frac_of_peak_at_held_clock"], flush=True)

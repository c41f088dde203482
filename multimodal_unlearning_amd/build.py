"""Build libmmu_hip.so (gfx950 only) in-tree with hipcc.  `python -m multimodal_unlearning_amd.build`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmmu_hip.so")
SOURCES = ["gemm.hip", "attention.hip", "attention_dkdv4.hip", "elementwise.hip", "loss.hip", "optim.hip"]
# per-file extra flags (tools/build_variant.py honours the table too).  The attention kernels read their S / dP / dS
# accumulators with VALU right after the MFMA chains: with the MFMA results selected into VGPRs the AGPR -> VGPR copies
# go away (forward -3 %, backward -3 % in a same-process A/B, outputs bit-identical)
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
EXTRA_FLAGS: dict[str, list[str]] = {"attention.hip": _VGPR_FORM, "attention_dkdv4.hip": _VGPR_FORM}
HEADERS = ["common.h", "attention_common.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-variable"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src, *hdrs]):
            jobs.append([HIPCC, *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

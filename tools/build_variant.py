"""Build an A/B variant of libmmu_hip.so with extra -D flags into multimodal_unlearning_amd/csrc/variants/ (select it with
MMU_HIP_LIB=<path>).  Usage: python tools/build_variant.py NAME -DFOO=1 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_unlearning_amd import build as B

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(B.CSRC, "variants")
os.makedirs(out, exist_ok=True)
objs = []
for s in B.SOURCES:
    o = os.path.join(out, f"{name}_{s.replace('.hip', '.o')}")
    subprocess.run([B.HIPCC, *B.FLAGS, *B.EXTRA_FLAGS.get(s, []), *flags, "-c", os.path.join(B.CSRC, s), "-o", o], check=True)
    objs.append(o)
lib = os.path.join(out, f"libmmu_hip_{name}.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
for o in objs:
    os.remove(o)
print(lib)

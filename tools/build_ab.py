"""Build an experimental variant of the library for A/B runs (tools only; never loaded by the product):
   python tools/build_ab.py <name> <source.hip> [-DFLAG ...]  ->  tools/_ab/libcdseg_hip_<name>.so
The named source is recompiled with the extra flags and linked with the current objects of everything else."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import build
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
build.build_library(verbose=False)
bdir = os.path.join(ROOT, "cdsegnet_amd", "csrc", "_build")
out_dir = os.path.join(ROOT, "tools", "_ab")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, f"{name}_{src.replace('.hip', '.o')}")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", *flags, "-c",
                       os.path.join(ROOT, "cdsegnet_amd", "csrc", src), "-o", obj])
objs = [os.path.join(bdir, s.replace(".hip", ".o")) for s in build.SOURCES if s != src] + [obj]
lib = os.path.join(out_dir, f"libcdseg_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)

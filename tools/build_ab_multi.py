"""Experimental variant of the library with SEVERAL sources recompiled with extra flags (tools only; never loaded by the product):
   python tools/build_ab_multi.py <name> <a.hip,b.hip,...> [-DFLAG ...] [--f16]  ->  tools/_ab/libcdseg_hip[_f16]_<name>.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import build
name, srcs = sys.argv[1], sys.argv[2].split(",")
f16 = "--f16" in sys.argv
flags = [a for a in sys.argv[3:] if a != "--f16"] + (["-DCDSEG_LP_F16"] if f16 else [])
build.build_library(verbose=False)
bdir = os.path.join(ROOT, "cdsegnet_amd", "csrc", "_build_f16" if f16 else "_build")
out_dir = os.path.join(ROOT, "tools", "_ab")
os.makedirs(out_dir, exist_ok=True)
objs, procs = [], []
for s in build.SOURCES:
    if s in srcs:
        obj = os.path.join(out_dir, f"{name}{'_f16' if f16 else ''}_{s.replace('.hip', '.o')}")
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", *flags, "-c",
                                       os.path.join(ROOT, "cdsegnet_amd", "csrc", s), "-o", obj]))
        objs.append(obj)
    else:
        objs.append(os.path.join(bdir, s.replace(".hip", ".o")))
for p in procs:
    assert p.wait() == 0
lib = os.path.join(out_dir, f"libcdseg_hip{'_f16' if f16 else ''}_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)

"""Build libcdseg_hip.so and libcdseg_hip_f16.so (the C-ABI HIP library with bfloat16 / IEEE-half as its 16-bit type;
same sources, csrc/common.h) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with the
repo snapshot to the GPU box.  `python -m cdsegnet_amd.build` or
`__graft_entry__.build()` runs this.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcdseg_hip.so")
LIB_F16 = os.path.join(HERE, "libcdseg_hip_f16.so")
VARIANTS = [(LIB, "_build", []), (LIB_F16, "_build_f16", ["-DCDSEG_LP_F16"])]
SOURCES = ["serialize.hip", "plan.hip", "gemm.hip", "elementwise.hip", "attention.hip", "conv.hip", "stem.hip", "mlp.hip", "blockrr.hip", "deep.hip", "pool.hip", "runtime.hip", "testtime.hip", "train.hip", "prof.hip", "abi.hip"]
HEADERS = ["common.h", "curves.h", "prof.h", "deep.h", os.path.join("..", "..", "include", "cdseg.h")]
ARCH = "gfx950"


def _stale():
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    for lib, _, _ in VARIANTS:
        if not os.path.exists(lib):
            return True
        t = os.path.getmtime(lib)
        if any(os.path.getmtime(d) > t for d in deps):
            return True
    return False


def build_library(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs, links = [], []
    for lib, bdir, defs in VARIANTS:
        build_dir = os.path.join(HERE, "csrc", bdir)
        os.makedirs(build_dir, exist_ok=True)
        objs = []
        for src in SOURCES:
            obj = os.path.join(build_dir, src.replace(".hip", ".o"))
            objs.append(obj)
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++20", "-fPIC"] + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        links.append([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed on {src}")
    for cmd in links:
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))

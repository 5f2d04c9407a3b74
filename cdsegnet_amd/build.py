"""Build libcdseg_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

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
SOURCES = ["serialize.hip", "gemm.hip", "elementwise.hip", "attention.hip", "conv.hip", "stem.hip", "mlp.hip", "blockrr.hip", "runtime.hip", "testtime.hip", "train.hip", "prof.hip", "abi.hip"]
HEADERS = ["common.h", "curves.h", "prof.h", os.path.join("..", "..", "include", "cdseg.h")]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    build_dir = os.path.join(HERE, "csrc", "_build")
    os.makedirs(build_dir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(build_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++20", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))

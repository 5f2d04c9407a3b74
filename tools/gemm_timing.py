"""In-kernel stamps of gemm_dma_kernel (experimental build: python tools/build_ab.py gtime gemm.hip -DCDSEG_EXPERIMENTS
-DCDSEG_GEMM_TIMING): per block the time in the K loop and in the epilogue, and how many blocks were resident.
usage: python tools/gemm_timing.py M N K"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libcdseg_hip_gtime.so")
from cdsegnet_amd import ops

M, N, K = (int(a) for a in sys.argv[1:4])
dev = torch.device("cuda")
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev)
o = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
fn = lambda: ops.gemm(A, W, o, bias=b)  # noqa: E731
for _ in range(20):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
lib = _lib.load()
f = lib.cdseg_debug_gemm_timing
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
nblk = min(16384, (M + 127) // 128 * ((N + 127) // 128))
buf = np.zeros(nblk * 8, dtype=np.uint64)
assert f(buf.ctypes.data, buf.size) == 0
t = buf.reshape(nblk, 8).astype(np.float64)
t = t[t[:, 0] > 0]
base = t[:, 0].min()
span = (t[:, 1].max() - base) * 0.01
life = (t[:, 1] - t[:, 0]) * 0.01
print(f"gemm {M} x {N} x {K}: launch {1e3 * e0.elapsed_time(e1):.1f} us by events, first entry -> last exit {span:.1f} us, {len(t)} blocks of 512 threads")
print(f"block life: mean {life.mean():.2f} us (p5 {np.percentile(life, 5):.2f}, p95 {np.percentile(life, 95):.2f}); K loop {t[:, 2].mean():.0f} cycles, epilogue {t[:, 3].mean():.0f} cycles "
      f"(clock ~{(t[:, 2] + t[:, 3]).sum() / (life.sum() * 1e3):.2f} GHz)")
print(f"epilogue phases (cycles, thread 0): stage half 0 + barrier {t[:, 4].mean():.0f} | its items (LDS reads, stores) {t[:, 5].mean():.0f} | "
      f"barrier + stage half 1 + barrier {t[:, 6].mean():.0f} | its items {t[:, 7].mean():.0f}")
edges = np.arange(0, span + 0.5, 0.5)
alive = np.zeros(len(edges))
for a, bb in zip((t[:, 0] - base) * 0.01, (t[:, 1] - base) * 0.01):
    alive[int(a / 0.5):int(bb / 0.5) + 1] += 1
print(f"blocks resident: mean {alive.mean():.0f}, max {alive.max():.0f} (256 CUs)")
print("block start times (us), deciles:", np.round(np.percentile((t[:, 0] - base) * 0.01, [0, 10, 25, 50, 75, 90, 100]), 1))

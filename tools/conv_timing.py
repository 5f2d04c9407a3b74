#!/usr/bin/env python3
"""In-kernel phase stamps of the gathered-conv K loop (gemm_dma_kernel; experimental build:
python tools/build_ab.py gtime gemm.hip -DCDSEG_EXPERIMENTS -DCDSEG_GEMM_TIMING): per wave the cycles spent waiting for its
own DMA, in the step barrier, issuing the next step's DMAs, and in the fragment reads + MFMAs.
usage: python tools/conv_timing.py [level=3] [scenes=8]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libcdseg_hip_gtime.so")
from cdsegnet_amd import ops, synth
from tools.bench_gemm import time_op

level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda")
sc = synth.collate([synth.room_scene(i, 120000) for i in range(scenes)])
grid = torch.as_tensor(sc["grid_coord"]).to(dev).int().contiguous()
offs = np.concatenate([[0], sc["offset"]])
batch = torch.as_tensor(np.repeat(np.arange(scenes), np.diff(offs))).to(dev).int().contiguous()
depth = int(grid.max().item()).bit_length()
code = ops.encode4(grid, batch, depth)
zs, perm = ops.sort_pairs(code[0].contiguous())
gz, bz = ops.gather_rows(grid, perm), ops.gather_rows(batch, perm)
code4 = ops.encode4(gz, bz, depth)
n, d = len(grid), depth
for lvl in range(level):
    cl, seg, cnt = ops.pool_level(zs, 3)
    m = int(cnt.item())
    gz, bz, code4 = ops.pool_gather(seg, m, n, 1, gz, bz, code4)
    zs, n, d = code4[0].contiguous(), m, d - 1
c = [32, 64, 128, 256, 512][level]
nbr = ops.nbr_table(zs, gz, bz, d, 3, True)
x = torch.randn(n, c, device=dev).to(torch.bfloat16)
w = (torch.randn(c, 27 * c, device=dev) / (27 * c) ** 0.5).to(torch.bfloat16)
b = torch.randn(c, device=dev)
o = torch.empty(n, c, dtype=torch.bfloat16, device=dev)
us = time_op(lambda: ops.gemm(x, w, o, bias=b, nbr=nbr, kvol=27, nbr_kmajor=True), 10)
lib = _lib.load()
f = lib.cdseg_debug_gemm_ktiming
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(4096 * 16 * 8, dtype=np.uint64)
assert f(buf.ctypes.data, buf.size) == 0
t = buf.reshape(4096, 16, 8).astype(np.float64)
t = t[t[:, :, 7] > 0]
steps = t[:, 4]
tot = t[:, 0] + t[:, 1] + t[:, 2] + t[:, 3]
print(f"conv level {level} n={n} C={c}: {us:.1f} us/launch (timing build); {len(t)} waves stamped, steps per block mean {steps.mean():.1f}, "
      f"live offsets mean {t[:, 6].mean():.1f}")
print(f"cycles per K step and wave: wait for own DMA {np.mean(t[:, 0] / steps):.0f} | barrier {np.mean(t[:, 1] / steps):.0f} | "
      f"issue next DMAs {np.mean(t[:, 2] / steps):.0f} | fragment reads + MFMAs {np.mean(t[:, 3] / steps):.0f} | total {np.mean(tot / steps):.0f}; "
      f"prologue {t[:, 5].mean():.0f} cycles")

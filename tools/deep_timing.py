#!/usr/bin/env python3
"""Phase stamps of the deep-stage head / tail kernels (experimental build:
python tools/build_ab.py dtime deep.hip -DCDSEG_EXPERIMENTS -DCDSEG_DEEP_TIMING): cycles of wave 0 between the phase
boundaries, mean over the stamped blocks.  usage: python tools/deep_timing.py [scenes=8]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libcdseg_hip_dtime.so")
from cdsegnet_amd import ops
from tools.bench_gemm import time_op

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev, bf = torch.device("cuda"), torch.bfloat16
lib = _lib.load()
f = lib.cdseg_debug_deep_timing
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_size_t]


def stamps(nblk):
    buf = np.zeros(2048 * 16, dtype=np.uint64)
    assert f(buf.ctypes.data, buf.size) == 0
    return buf.reshape(2048, 16)[:min(nblk, 2048)].astype(np.float64)


for n, C in ((14293 * scenes, 128), (3364 * scenes, 256)):
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    y, o = r(n, C).to(bf), r(n, C).to(bf)
    wl, wq, wp = (r(C, C) / C ** 0.5).to(bf), (r(3 * C, C) / C ** 0.5).to(bf), (r(C, C) / C ** 0.5).to(bf)
    w1, w2 = (r(4 * C, C) / C ** 0.5).to(bf), (r(C, 4 * C) / (4 * C) ** 0.5).to(bf)
    bl, bq, bp, b1, b2 = r(C), r(3 * C), r(C), r(4 * C), r(C)
    g1, e1, g2, e2 = r(C), r(C), r(C), r(C)
    x, xc, qkv = r(n, C), torch.empty(n, C, dtype=bf, device=dev), torch.empty(n, 3 * C, dtype=bf, device=dev)
    himg, timg = ops.block_rr_pack(C, wl, wq, wp, w1, w2)
    bm = 128 if (n + 127) // 128 >= 160 else 32
    nblk = (n + bm - 1) // bm
    us = time_op(lambda: ops.cpe_head_rr(y, himg, bl, (g1, e1), x, None, (g2, e2), bq, qkv), 10)
    t = stamps(nblk)
    d = np.diff(t[:, :13], axis=1).mean(0)
    print(f"head n={n} C={C} BM={bm}: {us:.1f} us; cycles: tile+params load {d[0]:.0f} | cpe product {d[1]:.0f} | x rows + LN_cpe stats {d[2]:.0f} | "
          f"x update + store {d[3]:.0f} | LN1 stats {d[4]:.0f} | prime + h write + barrier {d[5]:.0f} | q product {d[6]:.0f} store {d[7]:.0f} | "
          f"k product {d[8]:.0f} store {d[9]:.0f} | v product {d[10]:.0f} store {d[11]:.0f} | total {t[:, 12].mean() - t[:, 0].mean():.0f}")
    us = time_op(lambda: ops.attn_tail_rr(o, timg, bp, g1, e1, b1, b2, x, xc), 10)
    t = stamps(nblk)
    d = np.diff(t[:, :8], axis=1).mean(0)
    print(f"tail n={n} C={C} BM={bm}: {us:.1f} us; cycles: tile+params load {d[0]:.0f} | proj product {d[1]:.0f} | + x rows {d[2]:.0f} | LN2 stats {d[3]:.0f} | "
          f"h write + barrier {d[4]:.0f} | MLP chunks {d[5]:.0f} (fc1 products {t[:, 8].mean():.0f}, barrier + GELU + barrier {t[:, 9].mean():.0f}, "
          f"fc2 products {t[:, 10].mean():.0f}) | output stores {d[6]:.0f} | total {t[:, 7].mean() - t[:, 0].mean():.0f}")

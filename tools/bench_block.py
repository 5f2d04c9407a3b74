#!/usr/bin/env python3
"""Block head / tail kernels alone on stage-shaped problems: register-resident (blockrr.hip) vs 64-row-tile fused (mlp.hip).
usage: python tools/bench_block.py [scenes=8]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import _lib
if os.environ.get("CDSEG_AB_LIB"):  # A/B runs against another build of the library (tools only)
    _lib.LIB_PATH = os.path.abspath(os.environ["CDSEG_AB_LIB"])
from cdsegnet_amd import ops
from tools.bench_gemm import time_op

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev, bf = torch.device("cuda"), torch.bfloat16
for n, C in ((120000 * scenes, 32), (120000 * scenes, 64), (55818 * scenes, 64)):
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    y, o = r(n, C).to(bf), r(n, C).to(bf)
    wl, wq, wp = (r(C, C) / C ** 0.5).to(bf), (r(3 * C, C) / C ** 0.5).to(bf), (r(C, C) / C ** 0.5).to(bf)
    w1, w2 = (r(4 * C, C) / C ** 0.5).to(bf), (r(C, 4 * C) / (4 * C) ** 0.5).to(bf)
    bl, bq, bp, b1, b2 = r(C), r(3 * C), r(C), r(4 * C), r(C)
    g1, e1, g2, e2 = r(C), r(C), r(C), r(C)
    x, xc, qkv = r(n, C), torch.empty(n, C, dtype=bf, device=dev), torch.empty(n, 3 * C, dtype=bf, device=dev)
    himg, timg = ops.block_rr_pack(C, wl, wq, wp, w1, w2)
    hb, tb = n * (C * 2 + C * 8 + 3 * C * 2) / 1e6, n * (C * 2 + C * 8 + C * 2) / 1e6
    t_old = time_op(lambda: ops.cpe_head_fused(y, wl, bl, (g1, e1), x, None, (g2, e2), wq, bq, qkv), 10)
    t_new = time_op(lambda: ops.cpe_head_rr(y, himg, bl, (g1, e1), x, None, (g2, e2), bq, qkv), 10)
    print(f"head n={n} C={C}: fused {t_old:.1f} us, register-resident {t_new:.1f} us ({hb / t_new:.2f} TB/s on {hb:.0f} MB)")
    t_old = time_op(lambda: ops.attn_tail_fused(o, wp, bp, g1, e1, w1, b1, w2, b2, x, xc), 10)
    t_new = time_op(lambda: ops.attn_tail_rr(o, timg, bp, g1, e1, b1, b2, x, xc), 10)
    print(f"tail n={n} C={C}: fused {t_old:.1f} us, register-resident {t_new:.1f} us ({tb / t_new:.2f} TB/s on {tb:.0f} MB)")

# the C = 128 stage's MLP (weights too large for the register-resident tail: the 64 / 128-row tile kernel of mlp.hip)
n, C = 14293 * scenes, 128
r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
h, x, xc = r(n, C).to(bf), r(n, C), torch.empty(n, C, dtype=bf, device=dev)
w1, w2, b1, b2 = (r(4 * C, C) / C ** 0.5).to(bf), (r(C, 4 * C) / (4 * C) ** 0.5).to(bf), r(4 * C), r(C)
t = time_op(lambda: ops.mlp_fused(h, w1, b1, w2, b2, x, xc), 10)
mb = n * (C * 2 + C * 8 + C * 2) / 1e6
print(f"mlp n={n} C={C}: {t:.1f} us ({mb / t:.2f} TB/s on {mb:.0f} MB)")

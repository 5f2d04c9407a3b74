#!/usr/bin/env python3
"""Deep-stage Block head / tail (csrc/deep.hip, C = 128 / 256 / 512) against the separate GEMM launches they replace, on
stage-shaped problems.  usage: python tools/bench_deep.py [scenes=8] [f16|bf16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import _lib, ops
from tools.bench_gemm import time_op

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = sys.argv[2] if len(sys.argv) > 2 else "f16"
_lib.activate(variant)
dev = torch.device("cuda")
bf = torch.float16 if variant == "f16" else torch.bfloat16
shapes = ((14293 * scenes, 128), (3364 * scenes, 256), (778 * scenes, 128), (778 * scenes, 512), (778, 512), (3364, 256),
          (14293, 128))
for n, C in shapes:
    r = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
    y, o = r(n, C).to(bf), r(n, C).to(bf)
    wl, wq, wp = (r(C, C) / C ** 0.5).to(bf), (r(3 * C, C) / C ** 0.5).to(bf), (r(C, C) / C ** 0.5).to(bf)
    w1, w2 = (r(4 * C, C) / C ** 0.5).to(bf), (r(C, 4 * C) / (4 * C) ** 0.5).to(bf)
    bl, bq, bp, b1, b2 = r(C), r(3 * C), r(C), r(4 * C), r(C)
    g1, e1, g2, e2 = r(C), r(C), r(C), r(C)
    x, xc, qkv = r(n, C), torch.empty(n, C, dtype=bf, device=dev), torch.empty(n, 3 * C, dtype=bf, device=dev)
    h, u = torch.empty(n, C, dtype=bf, device=dev), torch.empty(n, 4 * C, dtype=bf, device=dev)
    himg, timg = ops.block_rr_pack(C, wl, wq, wp, w1, w2)

    def head_old():
        ops.gemm(y, wl, x, bias=bl, ln_pre=(g1, e1), res=x, ln_post=(g2, e2), ln_out=h)
        ops.gemm(h, wq, qkv, bias=bq)

    def tail_old():
        ops.gemm(o, wp, x, bias=bp, res=x, ln_post=(g1, e1), ln_out=h)
        if ops.mlp_fused_ok(h, 4 * C):
            ops.mlp_fused(h, w1, b1, w2, b2, x, xc)
        else:
            ops.gemm(h, w1, u, bias=b1, act=ops.ACT_GELU)
            ops.gemm(u, w2, x, bias=b2, res=x, out2=xc)

    hf, tf = 2.0 * n * 4 * C * C / 1e6, 2.0 * n * 9 * C * C / 1e6  # MFLOP
    t_old = time_op(head_old, 10)
    t_new = time_op(lambda: ops.cpe_head_rr(y, himg, bl, (g1, e1), x, None, (g2, e2), bq, qkv), 10)
    print(f"head n={n} C={C}: separate launches {t_old:.1f} us, fused {t_new:.1f} us ({hf / t_new:.0f} TFLOP/s)")
    t_old = time_op(tail_old, 10)
    t_new = time_op(lambda: ops.attn_tail_rr(o, timg, bp, g1, e1, b1, b2, x, xc), 10)
    print(f"tail n={n} C={C}: separate launches {t_old:.1f} us, fused {t_new:.1f} us ({tf / t_new:.0f} TFLOP/s)")
    if hasattr(ops, "cpe_head_rr2"):  # round 6: residual in / out in separate buffers -> few-row launches split a tile's stream
        x2, ws = torch.empty_like(x), torch.empty(4 * n * C * 4 + 64, dtype=torch.uint8, device=dev)
        t_h = time_op(lambda: ops.cpe_head_rr2(y, himg, bl, (g1, e1), x, x2, None, (g2, e2), bq, qkv), 10)
        t_t = time_op(lambda: ops.attn_tail_rr2(o, timg, bp, g1, e1, b1, b2, x2, x, xc, ws=ws), 10)
        print(f"rr2  n={n} C={C}: head {t_h:.1f} us, tail (+ reduce launch when split) {t_t:.1f} us")

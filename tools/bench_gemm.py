#!/usr/bin/env python3
"""Micro-benchmark of the step's GEMM / sparse-conv launch shapes (bf16), HIP-event timed, for tuning the
tile -> XCD map and the split-K policy (env: CDSEG_GEMM_XMODE, CDSEG_GEMM_SPLIT_TARGET, CDSEG_GEMM_SPLIT_MAX).
usage: python tools/bench_gemm.py [--iters 30]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import _lib  # noqa: E402
if os.environ.get("CDSEG_AB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CDSEG_AB_LIB"])
from cdsegnet_amd import ops, synth  # noqa: E402


def time_op(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--conv", action="store_true", help="include sparse-conv shapes (builds a 120k scene plan)")
    ap.add_argument("--kmajor", action="store_true", help="offset-major neighbour tables")
    ap.add_argument("--scenes", type=int, default=1, help="stage sizes of a batch of this many 120k-point scenes")
    args = ap.parse_args()
    dev = torch.device("cuda")
    bf = torch.bfloat16
    rows = []
    # (name, M, N, K, ln)
    stages = [(120000 * args.scenes, 32), (55818 * args.scenes, 64), (14293 * args.scenes, 128), (3364 * args.scenes, 256),
              (778 * args.scenes, 512)]
    shapes = []
    for n, c in stages:
        shapes += [(f"cpe-lin+LN n={n}", n, c, c, True), (f"qkv n={n}", n, 3 * c, c, False),
                   (f"proj+LN n={n}", n, c, c, True), (f"fc1 n={n}", n, 4 * c, c, False),
                   (f"fc2 n={n}", n, c, 4 * c, False)]
    for name, M, N, K, ln in shapes:
        A = torch.randn(M, K, device=dev).to(bf)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
        b = torch.randn(N, device=dev)
        x = torch.randn(M, N, device=dev)
        if ln:
            g1, b1 = torch.randn(N, device=dev), torch.randn(N, device=dev)
            h = torch.empty(M, N, dtype=bf, device=dev)
            fn = lambda: ops.gemm(A, W, x, bias=b, res=x, ln_post=(g1, b1), ln_out=h)  # noqa: E731
        elif N < K:
            fn = lambda: ops.gemm(A, W, x, bias=b, res=x)  # noqa: E731
        else:
            o = torch.empty(M, N, dtype=bf, device=dev)
            fn = lambda: ops.gemm(A, W, o, bias=b, act=ops.ACT_GELU)  # noqa: E731
        us = time_op(fn, args.iters)
        rows.append((name, M, N, K, us, 2.0 * M * N * K / us / 1e6))
    if args.conv:
        sc = synth.room_scene(0, 120000)
        grid = torch.as_tensor(sc["grid_coord"]).to(dev)
        batch = torch.zeros(len(grid), dtype=torch.int64, device=dev)
        depth = int(grid.max().item()).bit_length()
        g32, b32 = grid.int().contiguous(), batch.int().contiguous()
        code = ops.encode4(g32, b32, depth)
        zs, perm = ops.sort_pairs(code[0].contiguous())
        gz = ops.gather_rows(g32, perm)
        bz = ops.gather_rows(b32, perm)
        code4 = ops.encode4(gz, bz, depth)
        cur = (zs, gz, bz, code4, depth, len(grid))
        for lvl, (n_expect, c) in enumerate(stages):
            zs, gz, bz, code4, d, n = cur
            nbr = ops.nbr_table(zs, gz, bz, d, 3, args.kmajor)
            x = torch.randn(n, c, device=dev).to(bf)
            w = (torch.randn(c, 27 * c, device=dev) / (27 * c) ** 0.5).to(bf)
            b = torch.randn(c, device=dev)
            o = torch.empty(n, c, dtype=bf, device=dev)
            us = time_op(lambda: ops.gemm(x, w, o, bias=b, nbr=nbr, kvol=27, nbr_kmajor=args.kmajor), args.iters)
            rows.append((f"conv3 n={n}", n, c, 27 * c, us, 2.0 * n * c * 27 * c / us / 1e6))
            if lvl < 4:
                cl, seg, cnt = ops.pool_level(zs, 3)
                m = int(cnt.item())
                gc, bc, cc = ops.pool_gather(seg, m, n, 1, gz, bz, code4)
                cur = (cc[0].contiguous(), gc, bc, cc, d - 1, m)
    print(f"{'shape':28s} {'M':>7} {'N':>5} {'K':>6} {'us':>8} {'TFLOP/s':>8}")
    tot = 0.0
    for name, M, N, K, us, tf in rows:
        print(f"{name:28s} {M:7d} {N:5d} {K:6d} {us:8.2f} {tf:8.1f}")
        tot += us
    print(f"sum {tot:.1f} us   env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CDSEG_")))


if __name__ == "__main__":
    main()

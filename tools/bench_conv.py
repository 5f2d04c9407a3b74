#!/usr/bin/env python3
"""One sparse-conv (gathered-A GEMM) shape of the real scene plan, for rocprofv3 --pmc passes.
usage: python tools/bench_conv.py [level=0] [scenes=1] [iters=20]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import _lib
if os.environ.get("CDSEG_AB_LIB"):  # A/B runs against another build of the library (tools only)
    _lib.LIB_PATH = os.path.abspath(os.environ["CDSEG_AB_LIB"])
from cdsegnet_amd import ops, synth
from tools.bench_gemm import time_op

level = int(sys.argv[1]) if len(sys.argv) > 1 else 0
scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda")
torch.manual_seed(0)
sc = synth.collate([synth.room_scene(i, 120000) for i in range(scenes)])
grid = torch.as_tensor(sc["grid_coord"]).to(dev).int().contiguous()
offs = np.concatenate([[0], sc["offset"]])
batch = torch.as_tensor(np.repeat(np.arange(scenes), np.diff(offs))).to(dev).int().contiguous()
depth = int(grid.max().item()).bit_length()
code = ops.encode4(grid, batch, depth)
zs, perm = ops.sort_pairs(code[0].contiguous())
gz, bz = ops.gather_rows(grid, perm), ops.gather_rows(batch, perm)
code4 = ops.encode4(gz, bz, depth)
chans = [32, 64, 128, 256, 512]
n, d = len(grid), depth
for lvl in range(level):
    cl, seg, cnt = ops.pool_level(zs, 3)
    m = int(cnt.item())
    gz, bz, code4 = ops.pool_gather(seg, m, n, 1, gz, bz, code4)
    zs, n, d = code4[0].contiguous(), m, d - 1
c = chans[level]
nbr = ops.nbr_table(zs, gz, bz, d, 3, True)
occ = float((nbr >= 0).float().mean()) * 27
x = torch.randn(n, c, device=dev).to(torch.bfloat16)
w = (torch.randn(c, 27 * c, device=dev) / (27 * c) ** 0.5).to(torch.bfloat16)
b = torch.randn(c, device=dev)
o = torch.empty(n, c, dtype=torch.bfloat16, device=dev)
new_only = os.environ.get("CDSEG_BENCH_NEW_ONLY") is not None  # PMC passes: only the kernel under study
us = None if new_only else time_op(lambda: ops.gemm(x, w, o, bias=b, nbr=nbr, kvol=27, nbr_kmajor=True), iters)
if ops.subm_conv3_ok(x) and os.environ.get("CDSEG_BENCH_OLD_ONLY") is None:
    img = ops.subm_conv3_pack(w)
    o2 = torch.empty_like(o)
    us2 = time_op(lambda: ops.subm_conv3(x, img, b, nbr, o2), iters)
    comp = (n * c * 2 * 2 + n * 27 * 4 + 27 * c * c * 2) / 1e6  # features in + out, dense kernel map, weights
    print(f"conv level {level}: weight-stationary live-list kernel {us2:.1f} us/launch "
          f"({2.0 * n * occ * c * c / us2 / 1e6:.1f} TFLOP/s occupied; compulsory HBM bytes {comp:.1f} MB -> "
          f"{comp / us2:.2f} TB/s), checksum {float(o2.float().abs().sum()):.9e}" +
          ("" if us is None else f", max |new - gathered GEMM| = {(o2.float() - o.float()).abs().max().item():.3e}"))
if us is not None:  # (PMC passes run only the kernel under study: no gathered-GEMM line, no comparison against it)
    # spot check against a plain fp32 gather + matmul on 4096 sampled rows (same 16-bit operands)
    rows = torch.randperm(n, device=dev)[:4096]
    ref = b.repeat(len(rows), 1)
    wf = w.float().view(c, 27, c)
    for k in range(27):
        idx = nbr[k][rows].long()
        ref += torch.where((idx >= 0)[:, None], x.float()[idx.clamp(min=0)], torch.zeros(1, device=dev)) @ wf[:, k].t()
    err = (o[rows].float() - ref).abs().max().item()
    print(f"   spot check vs fp32 gather + matmul on 4096 rows: max |diff| {err:.3e} (outputs of magnitude {ref.abs().max().item():.2f}), "
          f"checksum {o.float().abs().sum().item():.6e}")
    print(f"conv level {level}: n={n} C={c} occupied neighbours/point={occ:.2f}: {us:.1f} us/launch, "
          f"{2.0 * n * occ * c * c / us / 1e6:.1f} TFLOP/s (occupied), gathered bytes {n * occ * c * 2 / 1e6:.1f} MB, "
          f"index bytes {n * 27 * 4 / 1e6:.1f} MB")

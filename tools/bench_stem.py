#!/usr/bin/env python3
"""The map-free 5x5x5 stem kernel alone on a collated batch of bench scenes.  usage: python tools/bench_stem.py [scenes=8]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import _lib
if os.environ.get("CDSEG_AB_LIB"):  # A/B runs against another build of the library (tools only)
    _lib.LIB_PATH = os.path.abspath(os.environ["CDSEG_AB_LIB"])
from cdsegnet_amd import ops, synth
from tools.bench_gemm import time_op

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
sc = synth.collate([synth.room_scene(i, 120000) for i in range(scenes)])
grid = torch.as_tensor(sc["grid_coord"]).to(dev).int().contiguous()
offs = np.concatenate([[0], sc["offset"]])
batch = torch.as_tensor(np.repeat(np.arange(scenes), np.diff(offs))).to(dev).int().contiguous()
n = len(grid)
depth = int(grid.max().item()).bit_length()
code = ops.encode4(grid, batch, depth)
zs, perm = ops.sort_pairs(code[0].contiguous())
g0, b0 = ops.gather_rows(grid, perm), ops.gather_rows(batch, perm)
code4 = ops.encode4(g0, b0, depth)
cluster, seg, cnt = ops.pool_level(zs, 3)
m = int(cnt.item())
g1, b1, c41 = ops.pool_gather(seg, m, n, 1, g0, b0, code4)
pn3 = ops.nbr_table(c41[0].contiguous(), g1, b1, depth - 1, 3, True)
cinfo = ops.child_info(zs, seg, m)
x = torch.randn(n, 8, device=dev).to(torch.bfloat16)
w = (torch.randn(32, 1000, device=dev) / 30).to(torch.bfloat16)
img = ops.stem5_pack(w)
s1, s2 = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev)
out = torch.empty(n, 32, dtype=torch.float32, device=dev)
out2 = torch.empty(n, 32, dtype=torch.bfloat16, device=dev)
t = time_op(lambda: ops.stem5(x, img, s1, s2, g0, cluster, pn3, cinfo, depth, out, out2), 10)
print(f"stem5 n={n} ({scenes} scenes), {m} parents: {t:.1f} us/launch, checksum {out.double().abs().sum().item():.6e}")

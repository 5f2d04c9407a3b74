#!/usr/bin/env python3
"""SerializedPooling feature path on stage-shaped problems: cdseg_gemm + cdseg_segment_max vs cdseg_pool_fused (csrc/pool.hip).
usage: python tools/bench_pool.py [scenes=8]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import ops
from tools.bench_gemm import time_op

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev, bf = torch.device("cuda"), torch.bfloat16
g = torch.Generator().manual_seed(0)
for name, n, m, cin, cout in (("n 0->1", 120000 * scenes, 55818 * scenes, 32, 64), ("c 0->2 (stride 4)", 120000 * scenes, 14293 * scenes, 32, 64),
                              ("n 1->2", 55818 * scenes, 14293 * scenes, 64, 128)):
    runs = torch.full((m,), n // m, dtype=torch.int64)
    runs[: n - int(runs.sum())] += 1
    runs = runs[torch.randperm(m, generator=g)]
    seg = torch.cat([torch.zeros(1, dtype=torch.int64), runs.cumsum(0)]).int().to(dev)
    x = torch.randn(n, cin, device=dev).to(bf)
    w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).to(bf)
    b, sc, sh = torch.randn(cout, device=dev), torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    img = ops.pool_fused_pack(w)
    y = torch.empty(n, cout, dtype=bf, device=dev)
    o1, o2 = torch.empty(m, cout, device=dev), torch.empty(m, cout, dtype=bf, device=dev)
    t_g = time_op(lambda: ops.gemm(x, w, y, bias=b), 10)
    t_s = time_op(lambda: ops.segment_max(y, seg, m, sc, sh, ops.ACT_GELU, o1, o2), 10)
    t_f = time_op(lambda: ops.pool_fused(x, img, b, seg, m, sc, sh, ops.ACT_GELU, o1, o2), 10)
    mb = (n * cin * 2 + m * cout * 6) / 1e6
    print(f"pooling {name}: n={n} -> m={m}, {cin} -> {cout}: gemm {t_g:.1f} + segment max {t_s:.1f} = {t_g + t_s:.1f} us; fused {t_f:.1f} us "
          f"({mb / t_f:.2f} TB/s on {mb:.0f} MB)")

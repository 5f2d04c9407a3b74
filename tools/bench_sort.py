#!/usr/bin/env python3
"""cdseg_sort_pairs (rocPRIM radix sort, 64-bit keys + int32 values) at the row counts of the plan.
usage: python tools/bench_sort.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import ops
from tools.bench_gemm import time_op

for n in (778, 3364, 14293, 55818, 120000, 200000, 262144, 300000, 444697, 864000, 1100000):
    keys = torch.randint(0, 1 << 33, (n,), dtype=torch.int64, device="cuda")
    t = time_op(lambda: ops.sort_pairs(keys, None, end_bit=34), 10)
    print(f"sort_pairs n={n}: {t:.1f} us")

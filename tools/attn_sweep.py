"""Same-process sweep of the serialized-attention kernel over library builds and schedule knobs (tools only).
usage: python tools/attn_sweep.py [--libs name=path,...] [--knobs "LEAD,TAIL1,TAIL2;..."] [--shapes "n:H:scenes;..."] [--f16]
  every library is opened with ctypes directly (the product loader is not involved); a library without cdseg_attention_ex
  (an older build) is called through cdseg_attention.  Knobs are environment variables read by -DCDSEG_EXPERIMENTS builds
  at every call (CDSEG_ATTN_LEAD / TAIL1 / TAIL2: blocks per XCD in the zones of the graded schedule, csrc/attention.hip).
Prints one line per (shape, library, knob setting): median / min microseconds per launch over 7 groups of `iters` launches, the
fraction of the 16-bit MFMA peak, and a checksum of the output (equal checksums = equal results).
The plans (serialization, slot plan) are built once per shape with the product library."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import _lib, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--libs", default="product=")
ap.add_argument("--knobs", default="0,0,0")
ap.add_argument("--shapes", default="864000:2:8;864000:4:8;402000:4:8;103000:8:8;27000:16:8;6200:32:8")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--f16", action="store_true", help="IEEE-half inputs (give f16 builds of the libraries)")
ap.add_argument("--flags", type=int, default=0, help="cdseg_attention_ex flags (1: q prescaled, 2: v bfloat16)")
ap.add_argument("--curve", type=int, default=2)
args = ap.parse_args()

dev = torch.device("cuda")
if args.f16:
    _lib.activate("f16")
tdtype = torch.float16 if args.f16 else torch.bfloat16


def open_lib(path):
    lib = ctypes.CDLL(path)
    sig = _lib.SIGNATURES
    if hasattr(lib, "cdseg_attention_ex"):
        fn = lib.cdseg_attention_ex
        fn.restype, fn.argtypes = sig["cdseg_attention_ex"]
        return fn, True
    fn = lib.cdseg_attention
    fn.restype, fn.argtypes = sig["cdseg_attention"]
    return fn, False


libs = []
for item in args.libs.split(","):
    name, _, path = item.partition("=")
    path = path or (_lib.LIB_PATH_F16 if args.f16 else _lib.LIB_PATH)
    libs.append((name, *open_lib(os.path.abspath(path))))
knobs = [tuple(int(v) for v in k.split(",")) for k in args.knobs.split(";")]


def build_case(n, H, scenes, curve):
    grids, batches = [], []
    for i in range(scenes):
        sc = synth.room_scene(i, n // scenes)
        grids.append(torch.as_tensor(sc["grid_coord"]))
        batches.append(torch.full((len(sc["grid_coord"]),), i, dtype=torch.int64))
    grid = torch.cat(grids).to(dev)
    batch = torch.cat(batches).to(dev)
    n = grid.shape[0]
    counts = torch.bincount(batch.cpu(), minlength=scenes).numpy()
    depth = int(ops.grid_max(grid).item()).bit_length()
    zs, perm0 = ops.sort_pairs(ops.encode(grid, batch, depth, "z"))
    g0, b0 = ops.plan_gather_grid(grid, perm0, zs, depth)
    code4 = ops.encode4(g0, b0, depth)
    _, order = ops.sort_pairs(code4[curve].contiguous())
    K = 1024
    pads = [(c + K - 1) // K * K if c > K else c for c in counts]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
    offs_pad = torch.tensor(np.concatenate([[0], np.cumsum(pads)]), dtype=torch.int32, device=dev)
    npad = int(sum(pads))
    gidx, widx = ops.pad_plan(order, offs, offs_pad, K, npad)
    starts = []
    for s, p in zip(np.concatenate([[0], np.cumsum(pads)])[:-1], pads):
        starts += list(range(int(s), int(s + p), K))
    ps = torch.tensor(starts + [npad], dtype=torch.int32, device=dev)
    C = 16 * H
    g = torch.Generator(device="cpu").manual_seed(n + H)
    qkv = torch.randn(n, 3 * C, generator=g).to(dev).to(tdtype)
    if args.flags & 2 and args.f16:  # v third as bfloat16 bits inside the half tensor
        qkv[:, 2 * C:] = qkv[:, 2 * C:].float().to(torch.bfloat16).view(torch.float16)
    out = torch.empty(n, C, dtype=tdtype, device=dev)
    lens = (ps[1:] - ps[:-1]).double()
    return dict(n=n, H=H, C=C, qkv=qkv, out=out, gidx=gidx, widx=widx, ps=ps, maxlen=int(lens.max().item()),
                flops=64.0 * H * float((lens * lens).sum()), units=(ps.numel() - 1) * H)


def launch(fn, is_ex, c):
    q, C = c["qkv"], c["C"]
    e = q.element_size()
    base = q.data_ptr()
    a = [base, base + C * e, base + 2 * C * e, 3 * C, 3 * C, 3 * C, c["gidx"].data_ptr(), c["gidx"].data_ptr(),
         c["widx"].data_ptr(), c["ps"].data_ptr(), c["ps"].numel() - 1, c["H"], c["maxlen"], 0.25, c["out"].data_ptr(), C, 1]
    if is_ex:
        a.append(args.flags)
    a.append(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    rc = fn(*a)
    assert rc == 0, rc


for shape in args.shapes.split(";"):
    n, H, scenes = (int(v) for v in shape.split(":"))
    c = build_case(n, H, scenes, args.curve)
    configs = [(name, fn, is_ex, kn) for name, fn, is_ex in libs for kn in (knobs if is_ex else [knobs[0]])]
    res = {i: [] for i in range(len(configs))}
    chk = {}

    def set_knobs(kn):
        for key, val in zip(("CDSEG_ATTN_LEAD", "CDSEG_ATTN_TAIL1", "CDSEG_ATTN_TAIL2"), kn):
            os.environ[key] = str(val)
        os.environ.pop("CDSEG_ATTN_QSPLIT", None)
        if len(kn) > 3 and kn[3] > 0:  # 4th value: uniform query split (0 = the library's own choice)
            os.environ["CDSEG_ATTN_QSPLIT"] = str(kn[3])
        os.environ["CDSEG_ATTN_W16"] = str(kn[4]) if len(kn) > 4 else "1"  # 5th: 16-wave blocks for one-block-per-CU launches
        os.environ["CDSEG_ATTN_W16_MIN_TILES"] = str(kn[5]) if len(kn) > 5 else "16"  # 6th: fewest query tiles per block for it

    set_knobs(configs[0][3])
    for _ in range(30):  # the chip settles on its sustained clock
        launch(configs[0][1], configs[0][2], c)
    torch.cuda.synchronize()
    for rnd in range(args.rounds):  # every config once per round, in order: slow drifts hit all configs alike
        for i, (name, fn, is_ex, kn) in enumerate(configs):
            set_knobs(kn)
            if rnd == 0:
                c["out"].zero_()
            for _ in range(3):
                launch(fn, is_ex, c)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                launch(fn, is_ex, c)
            e1.record()
            torch.cuda.synchronize()
            res[i].append(1e3 * e0.elapsed_time(e1) / args.iters)
            if rnd == 0:
                chk[i] = float(c["out"].float().abs().sum())
    for i, (name, fn, is_ex, kn) in enumerate(configs):
        us = sorted(res[i])
        med = us[len(us) // 2]
        print(f"n={c['n']} H={H} units={c['units']} lib={name} knobs={kn}: median {med:.1f} us (min {us[0]:.1f}) "
              f"frac {c['flops'] / med / 1e6 / 2500:.4f} checksum {chk[i]:.6e}", flush=True)

"""Same-process sweep of the wide-stage sparse conv (csrc/conv.hip, C = 32 / 64) over library builds (tools only).
usage: python tools/conv_sweep.py --libs base=,quad=tools/_ab/quad [--scenes 8] [--out gpurun_out/x.json]
  name=DIR: the directory holds libcdseg_hip.so and libcdseg_hip_f16.so (tools/build_variant.sh); empty DIR = the product pair.
Every library is opened with ctypes directly.  The kernel maps of stage 0 (C = 32) and stage 1 (C = 64) of `scenes` collated
bench scenes are built once with the product library; every library then packs the same weights and runs the same conv:
  * results are compared with the first library's BIT FOR BIT, in the bfloat16 and in the IEEE-half build;
  * timing: interleaved rounds (every library once per round), median microseconds per launch of the bfloat16 build.
Prints one line per (stage, library) and, with --out, writes {"us": {lib: {"32": t, "64": t}}, "equal": {lib: bool}}."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import _lib, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--libs", default="base=")
ap.add_argument("--scenes", type=int, default=8)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--out", default=None)
args = ap.parse_args()
dev = torch.device("cuda")


def open_pair(d):
    pair = {}
    for variant, fname, default in (("bf16", "libcdseg_hip.so", _lib.LIB_PATH), ("f16", "libcdseg_hip_f16.so", _lib.LIB_PATH_F16)):
        lib = ctypes.CDLL(os.path.abspath(os.path.join(d, fname)) if d else default)
        for fn in ("cdseg_subm_conv3_wimg_bytes", "cdseg_subm_conv3_pack", "cdseg_subm_conv3"):
            f = getattr(lib, fn)
            f.restype, f.argtypes = _lib.SIGNATURES[fn]
        pair[variant] = lib
    return pair


libs = []
for item in args.libs.split(","):
    name, _, d = item.partition("=")
    libs.append((name, open_pair(d)))

# ---- the kernel maps of stage 0 and stage 1 (as tools/bench_conv.py builds them)
sc = synth.collate([synth.room_scene(i, 120000) for i in range(args.scenes)])
grid = torch.as_tensor(sc["grid_coord"]).to(dev).int().contiguous()
offs = np.concatenate([[0], sc["offset"]])
batch = torch.as_tensor(np.repeat(np.arange(args.scenes), np.diff(offs))).to(dev).int().contiguous()
depth = int(grid.max().item()).bit_length()
code = ops.encode4(grid, batch, depth)
zs, perm = ops.sort_pairs(code[0].contiguous())
gz, bz = ops.gather_rows(grid, perm), ops.gather_rows(batch, perm)
code4 = ops.encode4(gz, bz, depth)
n, d = len(grid), depth
cases = {}
for level, c in ((0, 32), (1, 64)):
    if level:
        cl, seg, cnt = ops.pool_level(zs, 3)
        m = int(cnt.item())
        gz, bz, code4 = ops.pool_gather(seg, m, n, 1, gz, bz, code4)
        zs, n, d = code4[0].contiguous(), m, d - 1
    nbr = ops.nbr_table(zs, gz, bz, d, 3, True)
    g = torch.Generator(device="cpu").manual_seed(100 + c)
    cases[c] = dict(n=n, nbr=nbr, x=torch.randn(n, c, generator=g).to(dev), b=torch.randn(c, generator=g).to(dev),
                    w=(torch.randn(c, 27 * c, generator=g) / (27 * c) ** 0.5).to(dev))
stream = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def prepare(lib, case, c, dt):
    x, w = case["x"].to(dt).contiguous(), case["w"].to(dt).contiguous()
    img = torch.empty(lib.cdseg_subm_conv3_wimg_bytes(c), dtype=torch.uint8, device=dev)
    rc = lib.cdseg_subm_conv3_pack(w.data_ptr(), c, img.data_ptr(), stream())
    assert rc == 0, rc
    y = torch.zeros(case["n"], c, dtype=dt, device=dev)
    torch.cuda.synchronize()

    def run():
        rc = lib.cdseg_subm_conv3(x.data_ptr(), c, img.data_ptr(), case["b"].data_ptr(), case["nbr"].data_ptr(), case["n"], c,
                                  y.data_ptr(), c, stream())
        assert rc == 0, rc
    return run, y, (x, w, img)


result = {"us": {name: {} for name, _ in libs}, "equal": {name: True for name, _ in libs}, "scenes": args.scenes}
for c, case in cases.items():
    # bit-for-bit against the first library, both builds
    for variant, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        ref = None
        for name, pair in libs:
            run, y, keep = prepare(pair[variant], case, c, dt)
            run()
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
                assert torch.isfinite(ref.float()).all() and float(ref.float().abs().sum()) > 0
            elif not torch.equal(y.view(torch.int16), ref.view(torch.int16)):
                result["equal"][name] = False
                print(f"C={c} {variant} {name}: DIFFERS from {libs[0][0]} (max |d| {(y.float() - ref.float()).abs().max().item():.3e})")
    runs = [(name, *prepare(pair["bf16"], case, c, torch.bfloat16)) for name, pair in libs]
    for _ in range(30):  # the chip settles on its sustained clock
        runs[0][1]()
    torch.cuda.synchronize()
    times = {name: [] for name, _ in libs}
    for rnd in range(args.rounds):
        for name, run, y, keep in runs:
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[name].append(1e3 * e0.elapsed_time(e1) / args.iters)
    for name, _ in libs:
        t = sorted(times[name])
        result["us"][name][str(c)] = t[len(t) // 2]
        print(f"conv C={c} n={case['n']} [{name}]: median {t[len(t) // 2]:.1f} us/launch (min {t[0]:.1f}, max {t[-1]:.1f}), "
              f"bit-identical to {libs[0][0]}: {result['equal'][name]}")
if args.out:
    with open(args.out, "w") as f:
        json.dump(result, f)

"""Timings of the SURVEY 8(f) rows around the single-step path, same build, one GPU:
  f1  raw scan -> labels: CenterShift, NormalizeColor, 13 test-time augmentations, GridSample(mode="test") fragments,
      per-fragment CenterShift + Collect, inference of every fragment (inference_many), softmax vote, arg-max
  f2  multi-step inference (inference_ddim, MSAI) with the plan built once
  f3  evaluator: arg-max, exact 1-NN label transfer to the raw points, IoU counters
usage: python tools/bench_next_rows.py  ->  text report on stdout"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from cdsegnet_amd import configs, evaluate, synth, testtime  # noqa: E402
from cdsegnet_amd.param_init import fill_state_dict  # noqa: E402
from cdsegnet_amd.registry import build_model  # noqa: E402
import cdsegnet_amd.models  # noqa: E402,F401


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, out


def main():
    dev = torch.device("cuda")
    cfg = configs.cdsegnet_config("scannet")
    model = build_model(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
    model = model.cuda().eval()
    model.precision, model.noise_source = "bf16", "device"

    # a raw scan: every 2 cm voxel of a 120k-voxel room holds 1-4 raw points (ScanNet: ~2 per voxel)
    sc = synth.room_scene(3, 120000)
    rng = np.random.default_rng(11)
    rep = rng.integers(1, 5, len(sc["coord"]))
    src = np.repeat(np.arange(len(rep)), rep)
    coord = (sc["coord"][src] + rng.uniform(-0.008, 0.008, (len(src), 3))).astype(np.float32)
    color = ((sc["feat"][src, :3] + 1) * 127.5).astype(np.float32)
    normal = sc["feat"][src, 3:].astype(np.float32)
    segment = sc["segment"][src]
    c, col, nrm = (torch.as_tensor(a).to(dev) for a in (coord, color, normal))
    n_raw = len(src)

    print(f"raw scan: {n_raw} points, {len(rep)} occupied 2 cm voxels")
    for name, augs in (("no augmentation (1 pass)", testtime.SCANNET_TTA[:1]), ("13 test-time augmentations", testtime.SCANNET_TTA)):
        idxs, dicts = testtime.prepare_test_fragments(c, col, nrm, 0.02, augs)
        torch.cuda.synchronize()
        tp, _ = timed(lambda: testtime.prepare_test_fragments(c, col, nrm, 0.02, augs), 3)
        tt, (labels, _) = timed(lambda: testtime.segment_scene_tta(model, c, col, nrm, 0.02, 20, augs=augs, lanes=3), 2)
        pts = sum(int(d["offset_host"][0]) for d in dicts)
        print(f"f1 {name}: {len(dicts)} fragments, {pts} fragment points; transforms + GridSample + Collect "
              f"{1e3 * tp:.1f} ms; raw scan -> labels {1e3 * tt:.1f} ms = {n_raw / tt / 1e6:.2f} M raw points/s, "
              f"{pts / tt / 1e6:.1f} M fragment points/s through the model")

    one = {k: torch.as_tensor(sc[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset")}
    one["offset_host"] = [int(v) for v in sc["offset"]]
    t1, _ = timed(lambda: model.inference(dict(one), eval=False), 5)
    for step in (1, 4, 9):
        td, _ = timed(lambda: model.inference_ddim(dict(one), step=step, eval=False, mode="avg"), 3)
        print(f"f2 inference_ddim(step={step}, MSAI): {1e3 * td:.2f} ms per 120k-voxel scene = {step + 1} backbone passes, "
              f"{1e3 * td / (step + 1):.2f} ms per pass (single-step inference incl. its plan: {1e3 * t1:.2f} ms)")

    logits = model.inference(dict(one), eval=False)["seg_logits"]
    ev = dict(one)
    ev.update(origin_coord=c, origin_offset=torch.tensor([n_raw], dtype=torch.int32, device=dev),
              origin_segment=torch.as_tensor(segment).to(dev).int(), segment=torch.as_tensor(sc["segment"]).to(dev).int())
    te, counts = timed(lambda: evaluate.evaluate_scene(logits, ev, 20, reduce=False), 10)
    print(f"f3 evaluator (arg-max + 1-NN transfer of {len(rep)} voxel labels to {n_raw} raw points + IoU counters): "
          f"{1e3 * te:.2f} ms = {n_raw / te / 1e6:.0f} M raw points/s; points counted {int(counts[2].sum())}")


if __name__ == "__main__":
    main()

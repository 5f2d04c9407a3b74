#!/usr/bin/env python3
"""One full-width training step (forward under autograd, loss.backward(), AdamW) on a synthetic ScanNet-shaped batch, fp32,
on the HIP kernels (cdsegnet_amd/train_graph.py).  Not a BASELINE metric - the reference publishes no training throughput -
a first number for the training row of SURVEY 8(f4).
usage: python tools/bench_train_step.py [scenes=1] [points=120000] [steps=4] [dataset=scannet|scannet200|nuscenes]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 1
points = int(sys.argv[2]) if len(sys.argv) > 2 else 120000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dataset = sys.argv[4] if len(sys.argv) > 4 else "scannet"
dev = torch.device("cuda")
cfg = configs.cdsegnet_config(dataset)
cfg["criteria"] = [dict(type="MSELoss", loss_weight=1.0, ignore_index=-1, batch_sample_point=-1),
                   dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1),
                   dict(type="LovaszLoss", mode="multiclass", loss_weight=1.0, ignore_index=-1)]
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0), strict=True)
model = model.to(dev).train()
sc = synth.collate([(synth.lidar_scene(i, points) if dataset == "nuscenes" else synth.room_scene(i, points)) for i in range(scenes)])
inp = {k: torch.as_tensor(sc[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset")}
inp["segment"] = (torch.as_tensor(np.asarray(sc["segment"]).astype(np.int64)) % cfg["num_classes"]).to(dev)
n = inp["feat"].shape[0]
named = dict(model.named_parameters())
opt = torch.optim.AdamW([dict(params=[p for k, p in named.items() if "block" not in k], lr=0.002),
                         dict(params=[p for k, p in named.items() if "block" in k], lr=0.0002)], lr=0.002, weight_decay=0.05)
times, losses = [], []
for it in range(steps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = model(inp)["loss"]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    losses.append(float(loss.detach()))
    if it:
        times.append((t1 - t0, t2 - t1, t3 - t2))
t = np.median(np.array(times), axis=0) * 1e3
print(f"training step, {dataset}, full width, fp32, {scenes} scene(s), {n} points: forward {t[0]:.1f} ms, backward {t[1]:.1f} ms, "
      f"AdamW {t[2]:.1f} ms = {t.sum():.1f} ms/step = {n / t.sum() * 1e3 / 1e6:.2f} M points/s; peak memory "
      f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; loss over the steps {[round(v, 4) for v in losses]}")

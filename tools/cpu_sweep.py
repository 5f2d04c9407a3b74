#!/usr/bin/env python3
"""Thread-count sweep of the CPU oracle (bench.py's cpu_baseline leg): which torch thread count is fastest on this host.
usage: python tools/cpu_sweep.py [points=24000] [threads...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401
from oracle import model as OM

points = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
threads = [int(v) for v in sys.argv[2:]] or [8, 16, 32, 64, 128]
cfg = configs.cdsegnet_config("scannet")
sd = fill_state_dict(build_model(cfg).state_dict(), seed=0)
sc = synth.room_scene(100, points)
n = len(sc["coord"])
inp = {k: sc[k] for k in ("coord", "grid_coord", "feat", "offset")}
draws = OM.draw_rng(1, n, cfg["c_in_channels"])
print(f"host cores: {os.cpu_count()}, scene {n} points")
for t in threads:
    torch.set_num_threads(t)
    OM.inference(cfg["backbone"], sd, inp, draws, T=cfg["T"])  # warm-up
    ts = []
    for _ in range(2):
        t0 = time.time()
        OM.inference(cfg["backbone"], sd, inp, draws, T=cfg["T"])
        ts.append(time.time() - t0)
    print(f"threads={t}: {min(ts):.2f} s -> {n / min(ts):.0f} points/s", flush=True)

"""Throughput of one precision mode over (scenes per forward, lanes) settings, same process, interleaved repetitions.
usage: python tools/mode_sweep.py <precision> "<batch>x<lanes>,..." [reps] [n_scenes]   (run on the GPU box)"""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401
prec = sys.argv[1]
settings = [tuple(int(v) for v in s.split("x")) for s in sys.argv[2].split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nsc = int(sys.argv[4]) if len(sys.argv) > 4 else max(b * l for b, l in settings)
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg); model.load_state_dict(fill_state_dict(model.state_dict(), seed=0)); model = model.cuda().eval()
model.noise_source = "device"; model.precision = prec
sizes = [102750 + (i * 34500 // max(1, nsc - 1)) for i in range(nsc)]
dicts = []
for i, n in enumerate(sizes):
    sc = synth.room_scene(9000 + i, n)
    d = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
    d["offset_host"] = [int(v) for v in sc["offset"]]
    dicts.append(d)
for rep in range(reps):
    for b, l in settings:
        sub = dicts[:b * l]
        pts = sum(int(d["feat"].shape[0]) for d in sub)
        for _ in range(2):
            model.inference_many([dict(d) for d in sub], lanes=l, batch=b)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3):
            model.inference_many([dict(d) for d in sub], lanes=l, batch=b)
        torch.cuda.synchronize(); el = (time.perf_counter() - t) / 3
        print(f"{prec}: {b} scenes/forward x {l} lanes: {pts / el / 1e6:.2f} M points/s, {1e3 * el:.1f} ms per step of {b * l} scenes", flush=True)

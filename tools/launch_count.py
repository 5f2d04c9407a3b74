"""Steady-state device work items of ONE forward (after warm-up, weights resident): kernel launches, device copies, fills.
   python tools/launch_count.py [precision]            (run on the GPU box)
Counts come from torch.profiler around a single `inference` call: (a) one 120k scene (the paper protocol's unit),
(b) eight collated scenes (the throughput benchmark's unit).  Model load / Engine.prepare copies are NOT in the window."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import configs, synth  # noqa: E402
from cdsegnet_amd.models import collate_device  # noqa: E402
from cdsegnet_amd.param_init import fill_state_dict  # noqa: E402
from cdsegnet_amd.registry import build_model  # noqa: E402
import cdsegnet_amd.models  # noqa: F401,E402

precision = sys.argv[1] if len(sys.argv) > 1 else "fp16+head"
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
model = model.cuda().eval()
model.precision = precision
model.noise_source = "device"
dicts = []
for i in range(8):
    sc = synth.room_scene(i, 120000)
    d = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
    d["offset_host"] = [int(v) for v in sc["offset"]]
    dicts.append(d)


def count(make_input, label):
    for _ in range(3):
        model.inference(make_input(), eval=False)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model.inference(make_input(), eval=False)
        torch.cuda.synchronize()
    kinds, names = Counter(), Counter()
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            n = ev.name
            kind = "memcpy" if n.startswith("Memcpy") else "memset" if n.startswith("Memset") else "kernel"
            kinds[kind] += 1
            names[n[:90]] += 1
    cpu = Counter(ev.name for ev in prof.events() if ev.device_type != torch.autograd.DeviceType.CUDA and ev.name.startswith("hip"))
    print(f"== {label}: kernels {kinds['kernel']}, device copies {kinds['memcpy']}, fills {kinds['memset']}")
    print("   host HIP calls:", dict(cpu))
    for n, c in names.most_common(60):
        print(f"   {c:4d}  {n}")


count(lambda: dict(dicts[0]), "single scene (bs = 1)")
count(lambda: collate_device([dict(d) for d in dicts]), "8 collated scenes (collate included)")

"""The parity modes (precision fp32 / fp32x3) on 8 collated 120k scenes, a few forwards - run under
`rocprofv3 --kernel-trace --stats` to see where their time goes.  usage: python tools/parity_mode_profile.py [fp32x3|fp32] [scenes=8]"""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from cdsegnet_amd import _lib
if os.environ.get("CDSEG_AB_LIB_F16"):  # A/B runs against another IEEE-half build of the library (tools only)
    _lib.LIB_PATH_F16 = os.path.abspath(os.environ["CDSEG_AB_LIB_F16"])
from cdsegnet_amd import configs, synth
from cdsegnet_amd.models import collate_device
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401
precision = sys.argv[1] if len(sys.argv) > 1 else "fp32x3"
scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
model = model.cuda().eval()
model.noise_source = "device"
model.precision = precision
dicts = []
for i in range(scenes):
    sc = synth.room_scene(i, 120000)
    d = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
    d["offset_host"] = [int(v) for v in sc["offset"]]
    dicts.append(d)
fwd = collate_device([dict(d) for d in dicts])
for _ in range(2):
    model.inference(dict(fwd), eval=False)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(4):
    model.inference(dict(fwd), eval=False)
torch.cuda.synchronize()
n = sum(d["feat"].shape[0] for d in dicts)
ms = 1e3 * (time.perf_counter() - t) / 4
print(f"{precision}: {scenes} collated scenes ({n} points): {ms:.2f} ms per forward = {n / ms / 1e3:.2f} M points/s (one forward at a time)", flush=True)

import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg); model.load_state_dict(fill_state_dict(model.state_dict(), seed=0)); model = model.cuda().eval(); model.precision = "bf16"; model.noise_source = "device"
sc = synth.room_scene(0, 120000)
inp = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}; inp["offset_host"] = [int(v) for v in sc["offset"]]
eng = model.engine()
for fs in (None, 0, 1, 2, 3):
    eng.fork_stage = fs
    for _ in range(5): model.inference(dict(inp), eval=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): model.inference(dict(inp), eval=False)
    torch.cuda.synchronize(); print(f"queues={os.environ.get('GPU_MAX_HW_QUEUES')} fork={fs}: {1e3*(time.perf_counter()-t)/20:.2f} ms/scene", flush=True)

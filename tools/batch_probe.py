import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, "/root/repo")
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg); model.load_state_dict(fill_state_dict(model.state_dict(), seed=0)); model = model.cuda().eval(); model.precision = "bf16"; model.noise_source = "device"
scs = [synth.room_scene(i, 120000) for i in range(4)]
for B in (1, 2, 3, 4):
    sc = synth.collate(scs[:B])
    inp = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
    inp["offset_host"] = [int(v) for v in sc["offset"]]
    for lanes in (2, 3, 4):
        model.inference_many([dict(inp) for _ in range(8)], lanes=lanes); torch.cuda.synchronize()
        K = 24
        t = time.perf_counter(); model.inference_many([dict(inp) for _ in range(K)], lanes=lanes); torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / K
        print(f"batch={B} lanes={lanes}: {1e3*dt:.2f} ms/forward = {1e3*dt/B:.2f} ms/scene, {B*120000/dt/1e6:.1f} M pts/s", flush=True)

"""Which torch-level ops (and device copies) does one inference step still issue?  (run on the GPU box)"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
model = model.cuda().eval(); model.precision = "bf16"; model.noise_source = "device"
sc = synth.room_scene(0, 120000)
inp = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
inp["offset_host"] = [int(v) for v in sc["offset"]]
for _ in range(3): model.inference(dict(inp), eval=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    model.inference(dict(inp), eval=False)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=40, max_name_column_width=60, max_src_column_width=110))

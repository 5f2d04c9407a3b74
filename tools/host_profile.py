"""Where does the host time of the inference loop go?  (run on the GPU box)
usage: python tools/host_profile.py [lanes]"""
import cProfile, pstats, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
model = model.cuda().eval(); model.precision = "bf16"; model.noise_source = "device"
sc = synth.room_scene(0, 120000)
inp = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
inp["offset_host"] = [int(v) for v in sc["offset"]]
model.inference_many([dict(inp) for _ in range(9)], lanes=lanes)
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
model.inference_many([dict(inp) for _ in range(K)], lanes=lanes)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"lanes={lanes}: host loop {1e3*(t1-t0)/K:.2f} ms/scene, GPU done {1e3*(t2-t0)/K:.2f} ms/scene")
pr = cProfile.Profile(); pr.enable()
model.inference_many([dict(inp) for _ in range(K)], lanes=lanes)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14); st.sort_stats("cumtime").print_stats(45)

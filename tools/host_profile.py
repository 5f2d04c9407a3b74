"""Where does the host time of one inference step go?  (run on the GPU box)"""
import cProfile, pstats, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
model = model.cuda().eval(); model.precision = "bf16"; model.noise_source = "device"
sc = synth.room_scene(0, 120000)
inp = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
inp["offset_host"] = [int(v) for v in sc["offset"]]
for _ in range(3): model.inference(dict(inp), eval=False)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); model.inference(dict(inp), eval=False); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host returns after {1e3*(t1-t0):.2f} ms, GPU done after {1e3*(t2-t0):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): model.inference(dict(inp), eval=False)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)

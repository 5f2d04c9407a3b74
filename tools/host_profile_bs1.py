"""bs = 1 (one inference(dict) per scene, the reference's protocol): where does the HOST time of a scene go?
cProfile over 30 scenes (its overhead inflates everything ~2x: read the ratios), then build_plan's own phases by wall clock
(no profiler).  usage: python tools/host_profile_bs1.py   (run on the GPU box)"""
import cProfile, pstats, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
model = model.cuda().eval(); model.noise_source = "device"
sc = synth.room_scene(0, 120000)
inp = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
for _ in range(10):
    model.inference(dict(inp), eval=False)
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
host = 0.0
for _ in range(K):
    a = time.perf_counter()
    model.inference(dict(inp), eval=False)
    host += time.perf_counter() - a
    torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"bs=1 (sync per scene): {1e3 * (t1 - t0) / K:.3f} ms per scene, of which inference() held the host {1e3 * host / K:.3f} ms")
eng = model.engine()
orig = eng.build_plan
acc = [0.0, 0]
def timed(*a, **k):
    t = time.perf_counter()
    r = orig(*a, **k)
    acc[0] += time.perf_counter() - t; acc[1] += 1
    return r
eng.build_plan = timed
for _ in range(K):
    model.inference(dict(inp), eval=False)
    torch.cuda.synchronize()
print(f"build_plan: {1e3 * acc[0] / acc[1]:.3f} ms of host wall per scene (incl. its blocking read)")
eng.build_plan = orig
pr = cProfile.Profile(); pr.enable()
for _ in range(K):
    model.inference(dict(inp), eval=False)
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(25); st.sort_stats("cumtime").print_stats(70)

"""One 120k-point scene at a time (batch 1), 30 inferences: wall per inference.  Run under
`rocprofv3 --kernel-trace --stats` and divide the kernel total by 30 to see how much of the wall time is GPU work
(the bs = 1 latency is kernel time, not host time).  usage: python tools/single_scene_profile.py"""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
from cdsegnet_amd import _lib
if os.environ.get("CDSEG_AB_LIB"):  # A/B runs against an experimental bfloat16 build (tools only): implies bf16+head
    _lib.LIB_PATH = os.path.abspath(os.environ["CDSEG_AB_LIB"])
if os.environ.get("CDSEG_AB_ENGINE"):  # A/B of the host side: another engine.py in place of the package's (tools only)
    import importlib.util
    spec = importlib.util.spec_from_file_location("cdsegnet_amd.engine", os.environ["CDSEG_AB_ENGINE"])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cdsegnet_amd.engine"] = mod
    spec.loader.exec_module(mod)
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
model = model.cuda().eval()
model.noise_source = "device"  # (precision: the default, fp16+head)
if os.environ.get("CDSEG_AB_LIB"):
    model.precision = "bf16+head"
sc = synth.room_scene(0, 120000)
inp = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
inp["offset_host"] = [int(v) for v in sc["offset"]]
for _ in range(10):
    model.inference(dict(inp), eval=False)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    model.inference(dict(inp), eval=False)
torch.cuda.synchronize()
print(f"single scene, {len(sc['coord'])} points: {1e3 * (time.perf_counter() - t) / 20:.2f} ms wall per inference "
      f"(30 inferences in this process incl. 10 warm-up)", flush=True)
if os.environ.get("CDSEG_SYNC_EACH"):  # the reference's protocol: the reference's dict, one synchronisation per scene
    ref = {k: inp[k] for k in ("coord", "grid_coord", "feat", "offset")}
    ts = []
    for _ in range(60):
        torch.cuda.synchronize()
        t = time.perf_counter()
        model.inference(dict(ref), eval=False)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    ts.sort()
    print(f"  synchronised per scene, no offset_host hint: median {1e3 * ts[len(ts) // 2]:.3f} ms, min {1e3 * ts[0]:.3f} ms", flush=True)

"""bs = 1: the host's timeline through one inference() - every cdsegnet_amd.ops call with its enter time, its duration and the
Python time between it and the previous call (median over 20 scenes, by call index).  No profiler: two perf_counter reads per
call.  usage: python tools/host_timeline_bs1.py [calls to print]   (run on the GPU box)"""
import sys, time, os, types, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import configs, synth, ops
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
LOG = []
SKIP = {"_ptr", "_stream", "check", "dt", "_dp", "current_stream_id", "workspace", "_need_gpu", "bind_stream", "unbind_stream"}
def wrap(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        LOG.append((name, t, time.perf_counter()))
        return r
    return w
for name, fn in list(vars(ops).items()):
    if isinstance(fn, types.FunctionType) and name not in SKIP and not name.startswith("__"):
        setattr(ops, name, wrap(name, fn))
runs = []
for _ in range(20):
    LOG.clear()
    t0 = time.perf_counter()
    model.inference(dict(inp), eval=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    runs.append((t0, t1, t2, list(LOG)))
names = [e[0] for e in runs[-1][3]]
runs = [r for r in runs if [e[0] for e in r[3]] == names]
show = int(sys.argv[1]) if len(sys.argv) > 1 else 90
med = statistics.median
print(f"{len(runs)} scenes with the same call sequence ({len(names)} ops calls per scene); inference() holds the host "
      f"{1e3 * med(r[1] - r[0] for r in runs):.3f} ms, the device finishes {1e3 * med(r[2] - r[1] for r in runs):.3f} ms later")
print(f"{'#':>4} {'enter us':>9} {'python before':>14} {'call us':>8}  op")
for i, nm in enumerate(names[:show]):
    ent = med(1e6 * (r[3][i][1] - r[0]) for r in runs)
    dur = med(1e6 * (r[3][i][2] - r[3][i][1]) for r in runs)
    gap = med(1e6 * (r[3][i][1] - (r[3][i - 1][2] if i else r[0])) for r in runs)
    print(f"{i:4d} {ent:9.1f} {gap:14.1f} {dur:8.1f}  {nm}")
tot_call = med(sum(e[2] - e[1] for e in r[3]) for r in runs)
print(f"sum of ops calls {1e3 * tot_call:.3f} ms; Python between them {1e3 * (med(r[1] - r[0] for r in runs) - tot_call):.3f} ms")

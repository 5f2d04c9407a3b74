"""bs = 1: a call tree of ONE inference() on the host (sys.setprofile: Python and C calls with wall-clock enter / exit), pruned
to calls of at least MIN us.  The tracer roughly doubles the time of call-heavy code: read the proportions.
usage: python tools/host_trace_bs1.py [min_us] [max_depth]   (run on the GPU box)"""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401
MIN = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
MAXD = int(sys.argv[2]) if len(sys.argv) > 2 else 7
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
model = model.cuda().eval(); model.noise_source = "device"
sc = synth.room_scene(0, 120000)
inp = {k: torch.as_tensor(sc[k]).cuda() for k in ("coord", "grid_coord", "feat", "offset")}
for _ in range(10):
    model.inference(dict(inp), eval=False)
torch.cuda.synchronize()
events = []
pc = time.perf_counter
def prof(frame, event, arg):
    if event == "call":
        events.append((pc(), 0, frame.f_code.co_name + " @" + os.path.basename(frame.f_code.co_filename) + ":" + str(frame.f_lineno)))
    elif event == "return":
        events.append((pc(), 1, None))
    elif event == "c_call":
        events.append((pc(), 0, getattr(arg, "__qualname__", None) or getattr(arg, "__name__", str(arg))))
    elif event in ("c_return", "c_exception"):
        events.append((pc(), 1, None))
d = dict(inp)
sys.setprofile(prof)
model.inference(d, eval=False)
sys.setprofile(None)
torch.cuda.synchronize()
t0 = events[0][0]
stack, nodes = [], []
for t, kind, name in events:
    if kind == 0:
        node = [name, t, None, len(stack)]
        nodes.append(node)
        stack.append(node)
    elif stack:
        stack.pop()[2] = t
END = 1e6 * (float(sys.argv[3]) if len(sys.argv) > 3 else 0.9e-3)
print(f"calls of >= {MIN} us, depth <= {MAXD}, entered in the first {END:.0f} us (enter us, duration us)")
for name, a, b, depth in nodes:
    if b is None or depth > MAXD:
        continue
    if 1e6 * (b - a) >= MIN and 1e6 * (a - t0) < END:
        print(f"{1e6 * (a - t0):8.1f} {1e6 * (b - a):8.1f}  {'  ' * depth}{name}")

"""Error budget of the bf16 mode: which stages of the network the bf16-vs-fp32 logit error and
the arg-max disagreements come from, and what each exact-fp32 stage costs.  One bench scene; the reference is the
exact-fp32 HIP path on the same draws; `hi` = the stages run through the fp32 twin engine (Engine.hi).
usage: python tools/bf16_budget.py [points] [bf16|fp16] > profiles/r03_bf16_budget.txt   (r04_precision.txt: fp16)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import configs, synth
from cdsegnet_amd.param_init import fill_state_dict
from cdsegnet_amd.registry import build_model
import cdsegnet_amd.models  # noqa: F401

points = int(sys.argv[1]) if len(sys.argv) > 1 else 103000
LOWP = sys.argv[2] if len(sys.argv) > 2 else "bf16"  # the 16-bit trunk under study
dev = torch.device("cuda")
cfg = configs.cdsegnet_config("scannet")
model = build_model(cfg)
model.load_state_dict(fill_state_dict(model.state_dict(), seed=0), strict=True)
model = model.to(dev).eval()
sc = synth.room_scene(0, points)
d0 = {k: torch.as_tensor(sc[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset")}
n = len(sc["coord"])
gen = torch.Generator().manual_seed(54421566)
draws = dict(noise=torch.normal(0, 1, size=(n, cfg["c_in_channels"]), dtype=torch.float32, generator=gen),
             perms=[torch.randperm(4, generator=gen).tolist() for _ in range(8)])
model.noise_source = "torch_cpu"

def run(precision, hi=(), exact_attention=False):
    model.precision = precision
    eng = model.engine()
    eng.hi = frozenset(hi) if precision != "fp32" else frozenset()
    eng.exact_attention_core = exact_attention  # (binding path only: the native Block executor is switched off with it)
    if exact_attention:
        model._drop_engine()
        eng = model.engine()
        eng.use_native_blocks = False
        eng.hi = frozenset(hi)
        eng.exact_attention_core = True
    out = model.inference(dict(d0), eval=False, draws=dict(draws))["seg_logits"].clone()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        model.inference(dict(d0), eval=False, draws=dict(draws))
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return out, float(np.median(ts))

ref, t_ref = run("fp32")
top2 = ref.topk(2, dim=1).values
margin = (top2[:, 0] - top2[:, 1])
print(f"scene: {n} voxels, 20 classes, random-init weights; reference = exact-fp32 HIP path ({t_ref:.1f} ms per scene)")
print(f"fp32 logits: mean |logit| {float(ref.abs().mean()):.3f}; top-1 / top-2 margin: median {float(margin.median()):.4f}, "
      f"{100 * float((margin < 0.01).float().mean()):.2f} % of points below 0.01, {100 * float((margin < 0.03).float().mean()):.2f} % below 0.03")
ENC = [f"n_enc{s}" for s in range(5)]
DEC = [f"n_dec{s}" for s in range(4)]
CB = ["c_emb", "c_enc0", "c_enc1", "c_enc2"]
ALL = ["n_emb"] + ENC + CB + ["x"] + DEC + ["n_head"]
rows = [(f"pure {LOWP}", ())]
rows += [(f"fp32: {k}", (k,)) for k in ALL]
rows += [("fp32: head + n_dec0", ("n_head", "n_dec0")),
         ("fp32: whole n-decoder + head", tuple(DEC) + ("n_head",)),
         ("fp32: whole n-encoder + stem", ("n_emb",) + tuple(ENC)),
         ("fp32: c-branch + cross block", tuple(CB) + ("x",)),
         ("fp32: stages 0 (n_enc0, c_enc0, n_dec0, stems, head)", ("n_emb", "c_emb", "n_enc0", "c_enc0", "n_dec0", "n_head")),
         ("fp32: deep stages (n_enc2..4, c_enc1..2, x, n_dec2..3)", ("n_enc2", "n_enc3", "n_enc4", "c_enc1", "c_enc2", "x", "n_dec2", "n_dec3")),
         ("fp32: everything (sanity)", tuple(ALL))]
base_t = None
print(f"{'configuration':58s} {'arg-max agree':>13s} {'max |dlogit|':>13s} {'rms dlogit':>11s} {'ms/scene':>9s} {'cost':>7s}")
rows.append((f"{LOWP} trunk, attention core (softmax(q k^T) v) in fp32", "ATT"))
for name, hi in rows:
    out, t = run(LOWP, () if hi == "ATT" else hi, exact_attention=(hi == "ATT"))
    if base_t is None:
        base_t = t
    diff = (out - ref)
    print(f"{name:58s} {100 * float((out.argmax(1) == ref.argmax(1)).float().mean()):12.3f}% {float(diff.abs().max()):13.4e} "
          f"{float(diff.pow(2).mean().sqrt()):11.3e} {t:9.2f} {100 * (t / base_t - 1):+6.1f}%")

"""In-kernel phase timing of the bf16 attention kernel (experimental build with -DCDSEG_ATTN_TIMING):
   python tools/build_ab.py timing attention.hip -DCDSEG_ATTN_TIMING
   python tools/attn_timing.py [n_points] [heads] [curve]
Every wave stamps s_memrealtime (100 MHz) at entry / exit and s_memtime (shader cycles) around the staging and the
key loops; prints the clock the chip held, the share of a block's life spent staging / in the key loops, and how full
the chip was over the launch."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdsegnet_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_ab", "libcdseg_hip_timing.so")
from cdsegnet_amd import ops, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 960000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 2
curve = int(sys.argv[3]) if len(sys.argv) > 3 else 2
C = 16 * H
dev = torch.device("cuda")
sc = synth.room_scene(0, n)
grid = torch.as_tensor(sc["grid_coord"]).to(dev)
n = grid.shape[0]
batch = torch.zeros(n, dtype=torch.int64, device=dev)
depth = int(ops.grid_max(grid).item()).bit_length()
zs, perm0 = ops.sort_pairs(ops.encode(grid, batch, depth, "z"))
g0, b0 = ops.plan_gather_grid(grid, perm0, zs, depth)
code4 = ops.encode4(g0, b0, depth)
_, order = ops.sort_pairs(code4[curve].contiguous())
K = 1024
npad = (n + K - 1) // K * K
offs = torch.tensor([0, n], dtype=torch.int32, device=dev)
offs_pad = torch.tensor([0, npad], dtype=torch.int32, device=dev)
gidx, widx = ops.pad_plan(order, offs, offs_pad, K, npad)
ps = torch.arange(0, npad + 1, K, dtype=torch.int32, device=dev)
qkv = torch.randn(n, 3 * C, device=dev).to(torch.bfloat16)
out = torch.empty(n, C, dtype=torch.bfloat16, device=dev)
def run():
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], gidx, gidx, widx, ps, H, K, 0.25, out)
for _ in range(30): run()   # long warm-up: the chip settles on its sustained clock
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib = _lib.load()
form = int(os.environ.get("CDSEG_ATTN_FORM", "1"))  # honoured by experimental builds only
wpb = 16 if form == 1 else 8
nblk = 2048 if form == 1 else 4096
buf = np.zeros(nblk * wpb * 8, dtype=np.uint64)
fn = lib.cdseg_debug_attn_timing
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert fn(buf.ctypes.data, buf.size) == 0
t = buf.reshape(nblk, wpb, 8).astype(np.float64)
live = t[:, 0, 0] > 0
t = t[live]
rt0, rt1 = t[:, :, 0], t[:, :, 1]
base = rt0.min()
span_us = (rt1.max() - base) * 0.01
life_cyc = t[:, :, 4]
life_us = (rt1 - rt0) * 0.01
print(f"launch {1e3 * e0.elapsed_time(e1):.1f} us by events; first entry -> last exit {span_us:.1f} us; {len(t)} blocks")
print(f"clock held (wave cycles / wave wall time): {life_cyc.sum() / (life_us.sum() * 1e3):.3f} GHz")
print(f"wave life: mean {life_cyc.mean():.0f} cycles = {life_us.mean():.1f} us (p5 {np.percentile(life_us, 5):.1f}, p95 {np.percentile(life_us, 95):.1f})")
print(f"  staging (block form: entry -> barrier; persistent form: stage a patch-head two ahead + its DMAs + norms, publisher wave): mean {t[:, :, 2].mean():.0f} cycles = {100 * t[:, :, 2].sum() / life_cyc.sum():.1f} % of wave life")
print(f"  key loops:                  mean {t[:, :, 3].mean():.0f} cycles = {100 * t[:, :, 3].sum() / life_cyc.sum():.1f} % of wave life")
if form == 0:
    nt = np.maximum(t[:, :, 7], 1)
    print(f"  barrier -> first key loop (first tile claim + query rows): mean {t[:, :, 5].mean():.0f} cycles = {100 * t[:, :, 5].sum() / life_cyc.sum():.1f} % of wave life")
    print(f"  after a key loop -> next (loose check, normalise, store, next rows; stamped WITH a vmcnt(0)): mean {(t[:, :, 6] / nt).mean():.0f} cycles per tile, "
          f"{100 * t[:, :, 6].sum() / life_cyc.sum():.1f} % of wave life; query tiles per wave: mean {t[:, :, 7].mean():.2f}, min {t[:, :, 7].min():.0f}, max {t[:, :, 7].max():.0f}")
if form == 1:
    print(f"  spinning for an unpublished stage: {100 * t[:, :, 6].sum() / life_cyc.sum():.1f} % of wave life; "
          f"tasks (32-query tiles) per wave: mean {t[:, :, 5].mean():.1f}, min {t[:, :, 5].min():.0f}, max {t[:, :, 5].max():.0f}")
ntiles = (ps.numel() - 1) * H * 1024.0  # 32x32 tiles of the launch (L = 1024 patches)
print(f"  cycles per 32x32 tile and wave inside the key loops: {t[:, :, 3].sum() / ntiles:.1f}; "
      f"whole launch: {span_us * 1e3 * (life_cyc.sum() / (life_us.sum() * 1e3)) * 1024 / ntiles:.1f} cycles per tile and SIMD "
      f"(floor of the instruction mix: ~130, profiles/r03_ubench_pipes.txt)")
# residency over the launch: waves alive per 1 us bucket
edges = np.arange(0, span_us + 1.0, 1.0)
alive = np.zeros(len(edges))
s = ((rt0 - base) * 0.01).ravel(); e = ((rt1 - base) * 0.01).ravel()
for a, b in zip(s, e):
    alive[int(a):int(b) + 1] += 1
print(f"waves resident: mean {alive.mean():.0f} of 4096 slots ({100 * alive.mean() / 4096:.1f} %), "
      f"first 10 % of the launch {alive[:len(alive) // 10].mean():.0f}, last 10 % {alive[-len(alive) // 10:].mean():.0f}")
blk_start = (rt0.min(axis=1) - base) * 0.01
print("block start times (us), deciles:", np.round(np.percentile(blk_start, [0, 10, 25, 50, 75, 90, 100]), 1))
if form == 1:
    print("per wave slot (mean over blocks): life us | publishing cycles | key-loop cycles | spinning | tasks")
    for w in range(wpb):
        print(f"  wave {w:2d}: {life_us[:, w].mean():7.1f} | {t[:, w, 2].mean():9.0f} | {t[:, w, 3].mean():9.0f} | {t[:, w, 6].mean():9.0f} | {t[:, w, 5].mean():5.1f}")

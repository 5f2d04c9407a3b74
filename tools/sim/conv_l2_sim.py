"""CPU model: how often does the wide-stage sparse conv (csrc/conv.hip, C = 64) re-fetch a row of x from beyond the XCD's L2?
   python tools/sim/conv_l2_sim.py [scenes=8] [l2_mb=4] [blocks_per_xcd=32]
Builds the stage-1 kernel map of the bench's collated scenes in numpy (voxels of the 2x-pooled grid in z-order per scene: the
row order the engine uses), then replays the kernel's row-line requests (one 128-byte line per gathered row at C = 64)
through one LRU cache per XCD for several tile -> block maps:
  xcd-slices   the kernel's map: XCD x owns the x-th eighth of the tiles, its blocks walk that slice side by side
               (block b: tiles b, b + B, b + 2B ... of the slice; the B blocks of an XCD are taken to advance in step)
  round-robin  tile t on XCD t % 8 (what a plain blockIdx -> tile map gives: every XCD sees every eighth tile)
  xcd-narrow   as xcd-slices with B / 4 blocks per XCD in step (a quarter of the rows in flight)
The kernel map itself (27 x 4 bytes per row, read once) and the output rows (written once) stream through the same cache.
Prints requests, misses and the re-fetch factor of x (row-line misses / rows).  No GPU, no product code: numpy only."""
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cdsegnet_amd import synth  # noqa: E402

scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
l2_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
TILE = 256  # rows per block tile (8 waves x 32 rows)


def morton(g):
    """z-order code of (n, 3) non-negative ints (x lowest bit, like the engine's 'z' curve up to an axis permutation - the
    locality statistics do not depend on which axis comes first)."""
    code = np.zeros(len(g), dtype=np.int64)
    for b in range(16):
        for a in range(3):
            code |= ((g[:, a] >> b) & 1) << (3 * b + a)
    return code


rows, nbrs, base = [], [], 0
for s in range(scenes):
    g = np.unique(synth.room_scene(s, 120000)["grid_coord"] >> 1, axis=0)  # stage 1: the 2x-pooled grid
    g = g[np.argsort(morton(g), kind="stable")]
    key = (g[:, 0] << 40) | (g[:, 1] << 20) | g[:, 2]
    order = np.argsort(key)
    skey = key[order]
    nb = np.full((27, len(g)), -1, dtype=np.int64)
    o = 0
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                q = g + np.array([dx, dy, dz])
                ok = (q >= 0).all(1)
                qk = (q[:, 0] << 40) | (q[:, 1] << 20) | q[:, 2]
                pos = np.searchsorted(skey, qk)
                pos[pos >= len(skey)] = 0
                hit = ok & (skey[pos] == qk)
                nb[o, hit] = order[pos[hit]] + base
                o += 1
    nbrs.append(nb)
    base += len(g)
nbr = np.concatenate(nbrs, axis=1)
n = nbr.shape[1]
tiles = (n + TILE - 1) // TILE
occ = float((nbr >= 0).sum()) / n
print(f"stage 1 of {scenes} collated scenes: {n} rows, {occ:.2f} occupied neighbours per row, {tiles} tiles of {TILE} rows; "
      f"L2 {l2_mb} MB per XCD, {B} blocks per XCD")

LINES = int(l2_mb * (1 << 20) / 128)
X_BASE, MAP_BASE, Y_BASE = 0, 1 << 40, 1 << 41


def tile_requests(t):
    """Line ids a tile requests, in issue order: its slice of the kernel map, the gathered rows offset by offset, its output."""
    r0, r1 = t * TILE, min(n, (t + 1) * TILE)
    req = [MAP_BASE + (o * n + r0) * 4 // 128 + k for o in range(27) for k in range((r1 - r0) * 4 // 128 + 1)]
    blk = nbr[:, r0:r1]
    for w in range(0, r1 - r0, 32):  # a wave's 32 rows, the offsets any of them has
        sub = blk[:, w:w + 32]
        for o in range(27):
            v = sub[o]
            v = v[v >= 0]
            if len(v):
                req.extend((X_BASE + v).tolist())
    req.extend(range(Y_BASE + r0, Y_BASE + r1))
    return req


def replay(schedule):
    """schedule: per XCD the list of rounds, a round = the tiles its blocks work on side by side."""
    req_rows = miss_rows = req_all = miss_all = 0
    for rounds in schedule:
        cache = OrderedDict()
        for rnd in rounds:
            streams = [tile_requests(t) for t in rnd]
            # the blocks of a round advance together: interleave their requests in chunks of one wave-instruction (16 lines)
            pos = [0] * len(streams)
            live = True
            while live:
                live = False
                for i, sreq in enumerate(streams):
                    p = pos[i]
                    if p >= len(sreq):
                        continue
                    live = True
                    for line in sreq[p:p + 16]:
                        is_row = line < MAP_BASE
                        req_all += 1
                        req_rows += is_row
                        if line in cache:
                            cache.move_to_end(line)
                        else:
                            miss_all += 1
                            miss_rows += is_row
                            cache[line] = None
                            if len(cache) > LINES:
                                cache.popitem(last=False)
                    pos[i] = p + 16
    return req_rows, miss_rows, req_all, miss_all


def xcd_slices(blocks):
    per = (tiles + 7) // 8
    out = []
    for x in range(8):
        sl = list(range(x * per, min(tiles, (x + 1) * per)))
        out.append([sl[i:i + blocks] for i in range(0, len(sl), blocks)])
    return out


def round_robin(blocks):
    out = []
    for x in range(8):
        sl = list(range(x, tiles, 8))
        out.append([sl[i:i + blocks] for i in range(0, len(sl), blocks)])
    return out


for name, sched in (("xcd-slices", xcd_slices(B)), ("round-robin", round_robin(B)), ("xcd-narrow", xcd_slices(max(1, B // 4)))):
    rr, mr, ra, ma = replay(sched)
    print(f"{name:12s}: row-line requests {rr / 1e6:.2f} M, misses {mr / 1e6:.2f} M = {100.0 * mr / rr:.1f} % "
          f"(x fetched {mr / n:.2f} times); all requests {ra / 1e6:.2f} M, misses {ma / 1e6:.2f} M = {ma * 128 / 1e6:.0f} MB beyond L2")

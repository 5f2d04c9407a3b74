"""Seeded synthetic scenes shaped like the reference's datasets (no datasets ship here).

ScanNet-shape: an indoor room (floor / ceiling / walls + axis-aligned boxes) sampled
on surfaces and voxelised at 0.02 m, feat = [colour in U(-1,1)^3, unit normal];
nuScenes-shape: a 32-ring LiDAR sweep over a ground plane with boxes, voxelised at
0.05 m, feat = [coord, strength].  The dict layout is what Pointcept's ``Collect`` +
``collate_fn`` hand to ``model.inference`` (configs/scannet/CDSegNet.py:272-276,
datasets/utils.py:34-39): coord (N,3) f32, grid_coord (N,3) int64, feat (N,C) f32,
segment (N,) int64, offset (B,) int64 cumulative.  One point per voxel.
"""
import numpy as np


def voxelize(coord, voxel):
    """First point of every occupied voxel, in voxel-key order (stable, deterministic).
    Returns (index into coord, grid_coord int64 shifted to start at 0)."""
    g = np.floor(coord / voxel).astype(np.int64)
    g -= g.min(0)
    key = (g[:, 0] << 42) | (g[:, 1] << 21) | g[:, 2]
    _, idx = np.unique(key, return_index=True)
    return idx, g[idx]


def _room_faces(rng, dims, n_boxes):
    lx, ly, lz = dims
    boxes = [(np.zeros(3), np.array([lx, ly, lz]), True)]  # room shell, normals point inward
    for _ in range(n_boxes):
        size = rng.uniform([0.3, 0.3, 0.3], [max(0.4, lx / 3), max(0.4, ly / 3), max(0.4, lz / 2)])
        lo = rng.uniform([0.05, 0.05, 0.0], np.maximum(np.array([lx, ly, lz]) - size - 0.05, 0.06))
        lo[2] = 0.0
        boxes.append((lo, lo + size, False))
    faces = []
    for bi, (lo, hi, inward) in enumerate(boxes):
        for ax in range(3):
            for side in (0, 1):
                if bi > 0 and ax == 2 and side == 0:
                    continue  # box bottoms are hidden
                faces.append((lo, hi, inward, ax, side))
    return faces


def _dense_samples(rng, faces, step, voxel):
    """Jittered lattice (spacing `step`) on every face: dense, scan-like surfaces.  Planes are
    snapped to voxel centres so a surface is one cell thick."""
    cs, ns, ls = [], [], []
    for f, (lo, hi, inward, ax, side) in enumerate(faces):
        a1, a2 = (ax + 1) % 3, (ax + 2) % 3
        u = np.arange(lo[a1], hi[a1], step)
        v = np.arange(lo[a2], hi[a2], step)
        U, V = np.meshgrid(u, v, indexing="ij")
        p = np.empty((U.size, 3))
        p[:, a1] = U.ravel() + rng.uniform(0, step, U.size)
        p[:, a2] = V.ravel() + rng.uniform(0, step, U.size)
        p[:, ax] = (np.floor((hi[ax] if side else lo[ax]) / voxel) + 0.5) * voxel
        nrm = np.zeros((U.size, 3))
        s = 1.0 if side else -1.0
        nrm[:, ax] = -s if inward else s
        cs.append(p)
        ns.append(nrm)
        ls.append(np.full(U.size, f, dtype=np.int64))
    return np.concatenate(cs), np.concatenate(ns), np.concatenate(ls)


def room_scene(seed=0, target_points=120000, voxel=0.02, num_classes=20, n_boxes=10, exact=True, occupancy=0.55):
    """ScanNet-shaped scene: a room (4:3 floor plan, 2.6 m high) with box furniture; surfaces are
    one voxel thick and `occupancy` of their 2 cm cells are hit by the scan, which reproduces the
    stage sizes of a real ScanNet scene through the stride-2 poolings (the reference's in-code
    trace ptv3.py:1785-1794: 265838 -> 115601 -> 32435 -> 8428 -> 2196, ratios .435/.28/.26/.26).
    Sized so that it voxelises to ~target_points (exactly if exact)."""
    rng = np.random.default_rng(seed)
    lz = 2.6
    dense_target = target_points / occupancy
    # solve 2(lx*ly + (lx+ly)*lz) ~= 0.8 * dense_target * voxel^2 for lx with ly = 0.75 lx (boxes add the rest)
    area = 0.8 * dense_target * voxel * voxel
    lx = max(0.6, (-3.5 * lz + np.sqrt((3.5 * lz) ** 2 + 4 * 1.5 * area)) / (2 * 1.5))
    if lx < 1.5:
        lz = max(0.5, lx)
    scale = 1.0
    for _ in range(6):
        dims = (lx * scale, 0.75 * lx * scale, lz)
        frng = np.random.default_rng(seed + 7919)
        faces = _room_faces(frng, dims, n_boxes)
        coord, normal, label = _dense_samples(rng, faces, 0.6 * voxel, voxel)
        coord = coord + rng.normal(0, 0.1 * voxel, coord.shape)  # sensor noise
        idx, grid = voxelize(coord, voxel)
        if len(idx) >= dense_target and len(idx) <= 1.12 * dense_target:
            break
        scale *= (1.05 * dense_target / len(idx)) ** 0.5
    n_keep = target_points if exact else int(round(len(idx) * occupancy))
    if len(idx) > n_keep:
        keep = np.sort(rng.choice(len(idx), size=n_keep, replace=False))
        idx, grid = idx[keep], grid[keep]
        grid = grid - grid.min(0)
    # the loader hands points over in arbitrary (hash) order, not voxel-key order
    perm = rng.permutation(len(idx))
    idx, grid = idx[perm], grid[perm]
    color = rng.uniform(-1, 1, (len(idx), 3))
    feat = np.concatenate([color, normal[idx]], 1).astype(np.float32)
    return dict(
        coord=coord[idx].astype(np.float32),
        grid_coord=grid.astype(np.int64),
        feat=feat,
        segment=(label[idx] % num_classes).astype(np.int64),
        offset=np.array([len(idx)], dtype=np.int64),
    )


def lidar_scene(seed=0, target_points=40000, voxel=0.05, num_classes=16, rings=32):
    """nuScenes-shaped sweep: `rings` beams x azimuth steps on a ground plane + boxes."""
    rng = np.random.default_rng(seed)
    n_az = int(target_points * 1.35 / rings) + 8
    elev = np.deg2rad(np.linspace(-30.0, 10.0, rings))
    az = np.linspace(0, 2 * np.pi, n_az, endpoint=False)
    E, A = np.meshgrid(elev, az, indexing="ij")
    h = 1.8
    with np.errstate(divide="ignore"):
        r = np.where(E < -0.01, h / np.tan(-E), 60.0)
    r = np.clip(r, 2.0, 50.0)
    label = np.where(E < -0.01, 0, 1).astype(np.int64)
    # obstacles: angular sectors with a shorter range (vehicles / walls)
    for b in range(24):
        a0 = rng.uniform(0, 2 * np.pi)
        w = rng.uniform(0.05, 0.4)
        rr = rng.uniform(4.0, 35.0)
        m = (np.abs(((A - a0 + np.pi) % (2 * np.pi)) - np.pi) < w) & (r > rr) & (E > -0.35)
        r = np.where(m, rr, r)
        label = np.where(m, 2 + b % (num_classes - 2), label)
    r = r + rng.normal(0, 0.02, r.shape)
    x = r * np.cos(E) * np.cos(A)
    y = r * np.cos(E) * np.sin(A)
    z = r * np.sin(E) + h
    coord = np.stack([x, y, z], -1).reshape(-1, 3)
    label = label.reshape(-1)
    idx, grid = voxelize(coord, voxel)
    if len(idx) > target_points:
        keep = np.sort(rng.choice(len(idx), size=target_points, replace=False))
        idx, grid = idx[keep], grid[keep]
        grid = grid - grid.min(0)
    perm = rng.permutation(len(idx))
    idx, grid = idx[perm], grid[perm]
    strength = rng.uniform(0, 1, (len(idx), 1))
    c = coord[idx]
    feat = np.concatenate([c, strength], 1).astype(np.float32)
    return dict(
        coord=c.astype(np.float32),
        grid_coord=grid.astype(np.int64),
        feat=feat,
        segment=(label[idx] % num_classes).astype(np.int64),
        offset=np.array([len(idx)], dtype=np.int64),
    )


def perturb_scene(scene, seed=0, sigma=0.05, drop=0.5, voxel=0.02):
    """Robustness config: coord += N(0, sigma^2), drop a fraction, RE-VOXELISE."""
    rng = np.random.default_rng(seed)
    n = len(scene["coord"])
    keep = rng.random(n) >= drop
    coord = scene["coord"][keep].astype(np.float64) + rng.normal(0, sigma, (int(keep.sum()), 3))
    idx, grid = voxelize(coord, voxel)
    perm = rng.permutation(len(idx))
    idx, grid = idx[perm], grid[perm]
    src = np.nonzero(keep)[0][idx]
    return dict(
        coord=coord[idx].astype(np.float32),
        grid_coord=grid.astype(np.int64),
        feat=scene["feat"][src],
        segment=scene["segment"][src],
        offset=np.array([len(idx)], dtype=np.int64),
    )


def collate(scenes):
    """Concatenate scenes into one batch with cumulative offsets (datasets/utils.py:34-39)."""
    out = {}
    for k in ("coord", "grid_coord", "feat", "segment"):
        out[k] = np.concatenate([s[k] for s in scenes], 0)
    out["offset"] = np.cumsum([len(s["coord"]) for s in scenes]).astype(np.int64)
    return out

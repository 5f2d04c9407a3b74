"""Test-time pipeline around the model on the GPU: raw scan -> labels.

ref: configs/scannet/CDSegNet.py:253-398 (the ``test`` dataset block), pointcept/datasets/defaults.py:98-132
(prepare_test_data), pointcept/datasets/transform.py, pointcept/engines/test.py:197-279.  Per scene the reference
  1. transform:        CenterShift(apply_z=True), NormalizeColor                      (transform.py:142-155, :113-117)
  2. aug_transform:    13 test-time augmentations of the WHOLE scan: rotation about z by {0, 1/2, 1, 3/2} pi, each
                       plain / scaled 0.95 / scaled 1.05, plus one x-y flip            (:259-328)
  3. test_cfg.voxelize GridSample(grid_size, mode="test") on every augmented scan: ``count.max()`` fragments, fragment
                       i holds member ``i % count`` of every voxel                     (:821-897)
  4. post_transform:   CenterShift(apply_z=False) per fragment, ToTensor, Collect(keys=(coord, grid_coord, index),
                       feat_keys=(color, normal))                                      (:142-155, :27-50)
  5. tester:           model.inference on every fragment, pred[index] += softmax(logits), arg-max
                                                                                       (engines/test.py:197-279)
on CPU dataloader workers with numpy; here every step is a device kernel (csrc/testtime.hip) and the fragments of
all augmentations go through ``inference_many``.

Differences that cannot matter: voxels are keyed by a packed (x,y,z) integer instead of the FNV-1a hash (any injective
key forms the same groups); the sort is stable, so the members of a voxel are in original index order (numpy's default
argsort in the reference is unstable, i.e. platform dependent - SURVEY.md 8f): the per-point voxel coordinates are
compared exactly with the reference's (tests/golden/tta_pipeline.npz), fragment membership as sets per voxel.
Numerics follow numpy's promotions: a rotation multiplies float32 rows by a float64 matrix, so rotated coordinates
and normals are float64 from there on (GridSample divides float64), the flip stays float32.
"""
import math

import torch

from . import ops


def _rot_z(angle_pi):
    """rot_t of RandomRotateTargetAngle(axis="z") for angle = angle_pi * pi, with numpy's cos / sin VALUES
    (cos(pi/2) = 6.1e-17, not 0: the reference multiplies by exactly these doubles) - transform.py:272-279."""
    a = angle_pi * math.pi
    c, s = math.cos(a), math.sin(a)
    return [[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]


# configs/scannet/CDSegNet.py:278-398 (same list in scannet200 / nuscenes configs): (rotation angle / pi, scale, flip)
SCANNET_TTA = ([(a, None, False) for a in (0, 0.5, 1, 1.5)] + [(a, 0.95, False) for a in (0, 0.5, 1, 1.5)] +
               [(a, 1.05, False) for a in (0, 0.5, 1, 1.5)] + [(None, None, True)])


def apply_aug(coord, normal, aug):
    """One aug_transform entry on the (already centre-shifted) scan: (coord', normal')."""
    angle, scale, flip = aug
    if angle is not None:
        rot = _rot_z(angle)
        return ops.tta_apply(coord, rot=rot, scale=scale), (None if normal is None else ops.tta_apply(normal, rot=rot))
    if flip:
        return ops.tta_apply(coord, flip=True), (None if normal is None else ops.tta_apply(normal, flip=True))
    return coord, normal


def grid_sample_test(coord, grid_size):
    """-> dict(grid_coord int32 (N,3), idx_sort, seg_start, num_voxels, num_fragments)."""
    n = coord.shape[0]
    grid, key, _ = ops.voxelize_any(coord, grid_size)
    key_sorted, idx_sort = ops.sort_pairs(key, None, end_bit=63)
    _, seg_start, count = ops.pool_level(key_sorted, 0)
    m = int(count.item())
    frags = int(ops.max_run(seg_start, m).item())
    return dict(grid_coord=grid, idx_sort=idx_sort, seg_start=seg_start, num_voxels=m, num_fragments=frags, n=n)


def fragment(gs, i):
    """Indices (into the raw cloud) of fragment i: one point per voxel."""
    return ops.fragment_select(gs["idx_sort"], gs["seg_start"], gs["num_voxels"], i)


def _fragment_dicts(gs, coord, feat, max_fragments=None):
    nfrag = gs["num_fragments"] if max_fragments is None else min(gs["num_fragments"], max_fragments)
    idxs, dicts = [], []
    for i in range(nfrag):
        idx = fragment(gs, i)
        m = idx.numel()
        idxs.append(idx)
        # post_transform: CenterShift(apply_z=False) on the fragment's own coordinates, Collect
        c = ops.center_shift(ops.gather_rows(coord, idx), apply_z=False)
        dicts.append(dict(coord=c, grid_coord=ops.gather_rows(gs["grid_coord"], idx), feat=ops.gather_rows(feat, idx),
                          index=idx, offset=torch.tensor([m], dtype=torch.int64, device=coord.device), offset_host=[m]))
    return idxs, dicts


def _vote(model, n, num_classes, idxs, dicts, device, noise_level, lanes):
    if hasattr(model, "inference_many"):
        outs = model.inference_many(dicts, lanes=lanes, noise_level=noise_level)
    else:
        outs = [model.inference(d, eval=False, noise_level=noise_level) for d in dicts]
    ops.bind_stream()
    try:
        pred = torch.zeros((n, num_classes), dtype=torch.float32, device=device)
        for idx, o in zip(idxs, outs):
            ops.softmax_vote(o["seg_logits"], idx, pred)
        labels = ops.argmax_rows(pred)
    finally:
        ops.unbind_stream()
    return labels, pred


@torch.no_grad()
def segment_scene(model, coord, feat, grid_size, num_classes, noise_level=None, max_fragments=None, lanes=4):
    """Fragmented inference + softmax voting of one scan whose ``coord`` / ``feat`` are already normalised (steps 3-5,
    no augmentation).  coord (N,3) f32, feat (N,C) f32 on the GPU.  Returns (labels int32 (N,), pred (N, classes) f32).
    The fragments are independent scenes for the model, so they go through ``inference_many`` (up to ``lanes`` in
    flight); the votes are accumulated afterwards in fragment order, exactly like the reference's loop."""
    ops.bind_stream()
    try:
        gs = grid_sample_test(coord, grid_size)
        idxs, dicts = _fragment_dicts(gs, coord.float().contiguous(), feat.float().contiguous(), max_fragments)
    finally:
        ops.unbind_stream()
    return _vote(model, gs["n"], num_classes, idxs, dicts, coord.device, noise_level, lanes)


@torch.no_grad()
def prepare_test_fragments(coord, color, normal, grid_size, augs=SCANNET_TTA, max_fragments=None):
    """Steps 1-4 of the reference's test pipeline on the device: raw ``coord`` (N,3) f32 in metres, ``color`` (N,3) f32
    in 0..255, ``normal`` (N,3) f32 (or None: feat = colour only).  Returns (index lists, fragment dicts) over ALL
    augmentations in the reference's order (aug-major, fragment-minor), ready for model.inference / inference_many."""
    ops.bind_stream()
    try:
        coord0 = ops.center_shift(coord.float().contiguous(), apply_z=True)
        color0 = ops.div_add(color, 127.5, -1.0)
        idxs, dicts = [], []
        for aug in augs:
            c, nrm = apply_aug(coord0, normal, aug)
            feat = color0 if nrm is None else ops.collect_feat(color0, nrm)
            gs = grid_sample_test(c, grid_size)
            ii, dd = _fragment_dicts(gs, c, feat, max_fragments)
            idxs += ii
            dicts += dd
    finally:
        ops.unbind_stream()
    return idxs, dicts


@torch.no_grad()
def segment_scene_tta(model, coord, color, normal, grid_size, num_classes, augs=SCANNET_TTA, noise_level=None,
                      max_fragments=None, lanes=4):
    """The reference tester's per-scene work, raw scan -> labels (steps 1-5 with the config's 13 augmentations)."""
    idxs, dicts = prepare_test_fragments(coord, color, normal, grid_size, augs, max_fragments)
    return _vote(model, coord.shape[0], num_classes, idxs, dicts, coord.device, noise_level, lanes)

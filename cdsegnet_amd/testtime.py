"""Test-time pipeline around the model on the GPU.

ref: pointcept/datasets/transform.py:821-897 (GridSample, mode="test"): voxelise the raw scan, split it into
``count.max()`` fragments where fragment i holds member ``i % count`` of every voxel;
pointcept/engines/test.py:197-279: run the model on every fragment, accumulate
``pred[idx_part] += softmax(logits)``, arg-max.  The reference does the voxelisation on CPU workers
with numpy per scene; here it is a sort + scan + gather pipeline on the device.

Differences that cannot matter: voxels are keyed by a packed (x,y,z) integer instead of the FNV-1a hash
(any injective key forms the same groups); the sort is stable, so the members of a voxel are in original
index order (numpy's default argsort in the reference is unstable, i.e. platform dependent - SURVEY.md 8f).
Test-time augmentation (rotations / scales / flips of configs/scannet/CDSegNet.py:278-398) is a host-side
list of affine maps applied to ``coord`` before this pipeline and is not included.
"""
import torch

from . import ops


def grid_sample_test(coord, grid_size):
    """-> dict(grid_coord int32 (N,3), idx_sort, seg_start, num_voxels, num_fragments)."""
    n = coord.shape[0]
    grid, key, _ = ops.voxelize(coord, grid_size)
    key_sorted, idx_sort = ops.sort_pairs(key, None, end_bit=63)
    _, seg_start, count = ops.pool_level(key_sorted, 0)
    m = int(count.item())
    frags = int(ops.max_run(seg_start, m).item())
    return dict(grid_coord=grid, idx_sort=idx_sort, seg_start=seg_start, num_voxels=m, num_fragments=frags, n=n)


def fragment(gs, i):
    """Indices (into the raw cloud) of fragment i: one point per voxel."""
    return ops.fragment_select(gs["idx_sort"], gs["seg_start"], gs["num_voxels"], i)


@torch.no_grad()
def segment_scene(model, coord, feat, grid_size, num_classes, noise_level=None, max_fragments=None, lanes=4):
    """Fragmented inference + softmax voting of one raw scene (engines/test.py:181-279, bs = 1, no TTA).
    coord (N,3) f32, feat (N,C) f32 on the GPU.  Returns (labels int32 (N,), pred (N, num_classes) f32).
    The fragments are independent scenes for the model, so they go through ``inference_many`` (up to ``lanes`` in
    flight); the votes are accumulated afterwards in fragment order, exactly like the reference's loop."""
    ops.bind_stream()
    try:
        gs = grid_sample_test(coord, grid_size)
        n = gs["n"]
        pred = torch.zeros((n, num_classes), dtype=torch.float32, device=coord.device)
        nfrag = gs["num_fragments"] if max_fragments is None else min(gs["num_fragments"], max_fragments)
        coord_f, feat_f = coord.float().contiguous(), feat.float().contiguous()
        idxs, dicts = [], []
        for i in range(nfrag):
            idx = fragment(gs, i)
            m = idx.numel()
            idxs.append(idx)
            dicts.append(dict(coord=ops.gather_rows(coord_f, idx), grid_coord=ops.gather_rows(gs["grid_coord"], idx),
                              feat=ops.gather_rows(feat_f, idx),
                              offset=torch.tensor([m], dtype=torch.int64, device=coord.device), offset_host=[m]))
    finally:
        ops.unbind_stream()
    if hasattr(model, "inference_many"):
        outs = model.inference_many(dicts, lanes=lanes, noise_level=noise_level)
    else:
        outs = [model.inference(d, eval=False, noise_level=noise_level) for d in dicts]
    ops.bind_stream()
    try:
        for idx, o in zip(idxs, outs):
            ops.softmax_vote(o["seg_logits"], idx, pred)
        labels = ops.argmax_rows(pred)
    finally:
        ops.unbind_stream()
    return labels, pred

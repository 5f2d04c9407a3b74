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
def segment_scene(model, coord, feat, grid_size, num_classes, noise_level=None, max_fragments=None):
    """Fragmented inference + softmax voting of one raw scene (engines/test.py:181-279, bs = 1, no TTA).
    coord (N,3) f32, feat (N,C) f32 on the GPU.  Returns (labels int32 (N,), pred (N, num_classes) f32)."""
    ops.bind_stream()
    gs = grid_sample_test(coord, grid_size)
    n = gs["n"]
    pred = torch.zeros((n, num_classes), dtype=torch.float32, device=coord.device)
    nfrag = gs["num_fragments"] if max_fragments is None else min(gs["num_fragments"], max_fragments)
    for i in range(nfrag):
        ops.bind_stream()
        idx = fragment(gs, i)
        m = idx.numel()
        inp = dict(coord=ops.gather_rows(coord.float().contiguous(), idx),
                   grid_coord=ops.gather_rows(gs["grid_coord"], idx),
                   feat=ops.gather_rows(feat.float().contiguous(), idx),
                   offset=torch.tensor([m], dtype=torch.int64, device=coord.device), offset_host=[m])
        logits = model.inference(inp, eval=False, noise_level=noise_level)["seg_logits"]
        ops.bind_stream()
        ops.softmax_vote(logits, idx, pred)
    labels = ops.argmax_rows(pred)
    ops.unbind_stream()
    return labels, pred

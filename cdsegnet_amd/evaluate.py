"""Evaluator remap + metrics on the GPU.

ref: pointcept/engines/hooks/evaluator.py:128-146 - arg-max of the voxelised scene's logits, nearest-voxel label
transfer to the original points (``pointops.knn_query(1, coord, offset, origin_coord, origin_offset)``), then
``intersection_and_union_gpu`` (utils/misc.py:52-65) and the all-reduce of the three counters.
"""
import torch

from . import ops
from . import dist as cdist


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


@torch.no_grad()
def remap_labels(pred, input_dict, cell=None):
    """pred (N,) labels of the voxelised points -> labels of the original points (input_dict["origin_coord"],
    "origin_offset"), by exact 1-NN in ``coord``.  cell: search-grid cell size (default: 2.5 x the typical spacing
    estimated from the bounding box)."""
    coord = input_dict["coord"].float().contiguous()
    ocoord = input_dict["origin_coord"].float().contiguous()
    lo = coord.min(0).values
    if cell is None:
        ext = (coord.max(0).values - lo).clamp_min(1e-6)
        # surface-like scans: ~sqrt(n) points per side of the largest face
        cell = float(2.5 * (ext.sort().values[1:].prod() / max(1, coord.shape[0])).sqrt())
    idx = ops.knn1(coord, _i32(input_dict["offset"]), ocoord, _i32(input_dict["origin_offset"]), lo.cpu().tolist(), cell)
    return ops.gather_i32(_i32(pred).contiguous(), idx), idx


@torch.no_grad()
def evaluate_scene(seg_logits, input_dict, num_classes, ignore_index=-1, reduce=True):
    """The evaluator's per-scene step: returns the (3, K) int64 [intersection, union, target] counters (summed over
    ranks with one all-reduce when torch.distributed is initialised and ``reduce``)."""
    ops.bind_stream()
    try:
        pred = ops.argmax_rows(seg_logits.float().contiguous())
        if "origin_coord" in input_dict:
            _, idx = remap_labels(pred, input_dict)
            target = _i32(input_dict["origin_segment"]).contiguous()
            raw = ops.iou_counts(pred, target, num_classes, ignore_index, pred_idx=idx)
        else:
            raw = ops.iou_counts(pred, _i32(input_dict["segment"]).contiguous(), num_classes, ignore_index)
    finally:
        ops.unbind_stream()
    counts = torch.stack([raw[0], raw[1] + raw[2] - raw[0], raw[2]])
    return cdist.reduce_counts(counts) if reduce else counts

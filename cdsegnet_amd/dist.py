"""Scene-parallel inference across the GPUs of one node.

A scene (or a fragment of one) is a closed forward: no op spans scenes, so the path shards as
independent units - one process per GPU, no data-path collective.  The only traffic is
  * once:      broadcast of the weights from rank 0            (replaces the reference's DDP-ctor
               broadcast, ref: pointcept/engines/defaults.py:38, engines/test.py:62-66),
  * per eval:  all-reduce of the per-class intersection / union / target counters
               (replaces gloo gather_object of pickled records, ref: engines/test.py:374,
               utils/comm.py:169).
`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI); "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def rank_world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def shard_scenes(sizes, rank=None, world=None):
    """Longest-processing-time assignment of scenes (by point count) to ranks; returns the scene indices of
    `rank` in decreasing size.  Deterministic, every scene lands on exactly one rank."""
    if rank is None or world is None:
        rank, world = rank_world()
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += int(sizes[i])
        if r == rank:
            mine.append(i)
    return mine


def broadcast_model(model, src=0):
    """One flat broadcast of every state_dict entry (parameters + BN buffers) from `src`."""
    if not is_dist():
        return model
    sd = model.state_dict()
    tensors = list(sd.values())
    if not tensors:
        return model
    dev = tensors[0].device
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors]).to(dev)
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].reshape(t.shape).to(t.dtype))
            off += n
    if hasattr(model, "_drop_engine"):
        model._drop_engine()
    return model


def confusion_counts(pred, target, num_classes, ignore_index=-1):
    """Per-class intersection / union / target counts of one scene (ref: utils/misc.py:38-65), int64 (3, C)."""
    pred = pred.reshape(-1).clone()
    target = target.reshape(-1)
    pred[target == ignore_index] = ignore_index
    inter = pred[pred == target]
    ai = torch.bincount(inter[inter >= 0], minlength=num_classes)[:num_classes]
    ao = torch.bincount(pred[pred >= 0], minlength=num_classes)[:num_classes]
    at = torch.bincount(target[target >= 0], minlength=num_classes)[:num_classes]
    return torch.stack([ai, ao + at - ai, at]).to(torch.int64)


def reduce_counts(counts):
    """Sum the (3, C) counters over ranks (in place); every rank gets the totals."""
    if is_dist():
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def miou(counts):
    inter, union = counts[0].double(), counts[1].double()
    valid = union > 0
    return float((inter[valid] / union[valid]).mean()) if valid.any() else float("nan")

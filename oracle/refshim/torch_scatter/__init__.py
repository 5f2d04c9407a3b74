"""Stand-in for torch_scatter.segment_csr (reduce over CSR segments of dim 0)."""
import torch


def segment_csr(src, indptr, out=None, reduce="sum"):
    counts = (indptr[1:] - indptr[:-1]).long()
    m = counts.numel()
    seg = torch.repeat_interleave(torch.arange(m, device=src.device), counts)
    idx = seg.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    res = torch.zeros((m,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    red = {"sum": "sum", "mean": "mean", "max": "amax", "min": "amin"}[reduce]
    return res.scatter_reduce(0, idx, src, red, include_self=False)

"""Stand-in for the subset of spconv.pytorch the PTv3 path touches
(SparseConvTensor, SubMConv3d, modules.is_spconv_module).  Submanifold
cross-correlation, weight (out, k0, k1, k2, in), kernel axis a <-> indices[:, 1+a],
output sites = input sites.  See ../../README.md: conv arithmetic is OURS, so
parity is unpinned for it."""
import math

import torch
import torch.nn as nn

from . import modules  # noqa: F401


class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size, indice_dict=None):
        self.features = features
        self.indices = indices
        self.spatial_shape = spatial_shape
        self.batch_size = batch_size
        self.indice_dict = indice_dict if indice_dict is not None else {}

    def replace_feature(self, feature):
        return SparseConvTensor(feature, self.indices, self.spatial_shape, self.batch_size, self.indice_dict)


def _neighbour_table(indices, k):
    ind = indices.long()
    key = (ind[:, 0] << 48) | (ind[:, 1] << 32) | (ind[:, 2] << 16) | ind[:, 3]
    skey, srt = torch.sort(key)
    n = key.numel()
    r = k // 2
    cols = []
    for a in range(k):
        for b in range(k):
            for c in range(k):
                q = ind[:, 1:] + torch.tensor([a - r, b - r, c - r])
                ok = ((q >= 0) & (q < 65536)).all(1)
                qk = (ind[:, 0] << 48) | (q[:, 0] << 32) | (q[:, 1] << 16) | q[:, 2]
                pos = torch.searchsorted(skey, qk).clamp(max=n - 1)
                hit = ok & (skey[pos] == qk)
                cols.append(torch.where(hit, srt[pos], torch.full_like(pos, -1)))
    return torch.stack(cols, 1)


class SubMConv3d(modules.SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, algo=None):
        super().__init__()
        k = kernel_size
        self.kernel_size = k
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=math.sqrt(5))
        if bias:
            bound = 1 / math.sqrt(k * k * k * in_channels)
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        key = (self.indice_key, self.kernel_size)
        if self.indice_key is None or key not in x.indice_dict:
            tbl = _neighbour_table(x.indices, self.kernel_size)
            if self.indice_key is not None:
                x.indice_dict[key] = tbl
        else:
            tbl = x.indice_dict[key]
        cout = self.weight.shape[0]
        w = self.weight.reshape(cout, -1, self.weight.shape[-1])
        out = x.features.new_zeros(x.features.shape[0], cout)
        for kk in range(w.shape[1]):
            j = tbl[:, kk]
            m = j >= 0
            if m.any():
                out[m] += x.features[j[m]] @ w[:, kk, :].t()
        if self.bias is not None:
            out = out + self.bias
        return x.replace_feature(out)

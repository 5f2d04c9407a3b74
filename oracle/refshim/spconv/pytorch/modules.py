import torch.nn as nn


class SparseModule(nn.Module):
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)

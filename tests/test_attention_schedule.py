"""CPU: the block -> (patch, head, query slice) map of the serialized-attention launch (csrc/attention.hip: decode_block /
plan_zones, the graded schedule of round 5) through the library's host-only diagnostic cdseg_attention_schedule - the SAME
decode function the kernel runs.  Property: every (patch, head, slice) of a launch is covered by exactly one block id, the
slices of a patch-head share one slice count, and the blocks of an XCD come in launch order bulk -> half -> quarter slices.
No device work: runs in the GPU-less build container (the library loads without a GPU)."""
import ctypes

import numpy as np
import pytest

from cdsegnet_amd import _lib

BF16, F32 = _lib.BF16, _lib.F32


def schedule(P, H, max_len, dtype, variant="bf16"):
    lib = _lib.load(variant)
    n = lib.cdseg_attention_schedule(P, H, max_len, dtype, None, 0)
    assert n > 0
    tab = np.full((n, 4), -7, dtype=np.int32)
    assert lib.cdseg_attention_schedule(P, H, max_len, dtype, tab.ctypes.data_as(ctypes.c_void_p), n) == n
    return tab


# (patches, heads) of the benchmark's stages at 8, 1 and 24 collated scenes, deep stages, tiny launches
SHAPES = [(848, 2), (848, 4), (400, 4), (104, 8), (32, 16), (8, 32), (118, 2), (55, 4), (15, 8), (4, 16), (1, 32), (1, 2),
          (3, 1), (7, 1), (9, 2), (2544, 2), (1200, 4), (63, 2), (64, 2), (65, 2), (511, 1)]


@pytest.mark.parametrize("variant", ["bf16", "f16"])
@pytest.mark.parametrize("P,H", SHAPES)
@pytest.mark.parametrize("max_len", [1024, 700, 40])
def test_every_query_slice_is_scheduled_exactly_once(variant, P, H, max_len):
    for dtype in (BF16, F32):
        tab = schedule(P, H, max_len, dtype, variant)
        live = tab[tab[:, 0] >= 0]
        assert (tab[tab[:, 0] < 0] == -1).all()
        assert len(live) >= P * H
        patch, head, qslice, qsplit = live.T
        assert patch.max() == P - 1 and head.max() == H - 1 and patch.min() == 0 and head.min() == 0
        assert (qslice >= 0).all() and (qslice < qsplit).all()
        tile = 16 if dtype == F32 else 32
        nqt = (max_len + tile - 1) // tile
        assert (qsplit <= max(1, (nqt + 7) // 8)).all()  # a slice keeps at least one query tile per wave
        unit = patch.astype(np.int64) * H + head
        # one slice count per unit, all its slices present exactly once
        seen = {}
        for u, s, k in zip(unit, qslice, qsplit):
            seen.setdefault(int(u), (int(k), set()))
            assert seen[int(u)][0] == int(k)
            assert int(s) not in seen[int(u)][1]
            seen[int(u)][1].add(int(s))
        assert len(seen) == P * H
        assert all(len(sl) == k for k, sl in seen.values())


@pytest.mark.parametrize("P,H", [(848, 2), (400, 4), (104, 8)])
def test_xcd_runs_are_contiguous_and_end_with_finer_slices(P, H):
    tab = schedule(P, H, 1024, BF16)
    U = P * H
    for x in range(8):
        rows = tab[x::8]
        rows = rows[rows[:, 0] >= 0]
        unit = rows[:, 0].astype(np.int64) * H + rows[:, 1]
        assert unit.min() == U * x // 8 and unit.max() == U * (x + 1) // 8 - 1  # contiguous run of units per XCD
        assert (np.diff(unit) >= 0).all()                                       # in launch order
        split = rows[:, 3]
        k = len(split) // 2
        assert (np.diff(split[k:]) >= 0).all()  # towards the end of the launch the slices only get finer
    # the fp32 parity kernel keeps the uniform schedule
    t32 = schedule(P, H, 1024, F32)
    assert len(set(t32[t32[:, 0] >= 0][:, 3])) == 1

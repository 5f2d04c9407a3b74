"""Shared helpers for the parity tests (fixture loading, state dicts by name)."""
import json
import os

import numpy as np
import torch

from cdsegnet_amd.param_init import fill_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def fixture_cfg(fx):
    return json.loads(str(fx["cfg_json"]))


def fixture_state_dict(fx):
    """Regenerate the parameters of an e2e fixture by name and check the checksum."""
    keys = [str(k) for k in fx["sd_keys"]]
    shapes = {k: tuple(json.loads(str(s))) for k, s in zip(keys, fx["sd_shapes"])}
    sd = fill_state_dict(shapes, seed=int(fx["sd_seed"]))
    chk = sum(float(v.double().abs().sum()) for v in sd.values())
    assert abs(chk - float(fx["sd_checksum"])) <= 1e-6 * abs(chk), "param_init drifted from the golden fixtures"
    return sd


def fixture_draws(fx):
    d = dict(noise=torch.from_numpy(fx["noise"]), perms=[p for p in fx["perms"]])
    if "feat_noise" in fx.files:
        d["feat_noise"] = torch.from_numpy(fx["feat_noise"])
    return d


def fixture_input(fx):
    return dict(coord=fx["coord"], grid_coord=fx["grid_coord"], feat=fx["feat"], offset=fx["offset"])


def tiny_inputs(kind):
    """Degenerate scenes the reference's patch / pooling logic special-cases: fewer points than one patch, a batch whose
    elements differ by two orders of magnitude, points that pool down to ONE voxel per batch element before the last
    stage, a single point."""
    rng = np.random.default_rng(5)
    if kind == "one_point":
        grids = [np.array([[3, 1, 4]])]
    elif kind == "seven_and_many":
        grids = [rng.integers(0, 6, (7, 3)), rng.integers(0, 40, (600, 3))]
    elif kind == "collapses_early":  # all points inside one 4x4x4 block: one voxel after two poolings
        grids = [rng.integers(0, 4, (30, 3)), rng.integers(8, 12, (25, 3))]
    else:  # "small": 40 points, less than any patch size
        grids = [rng.integers(0, 12, (40, 3))]
    grids = [np.unique(g, axis=0) for g in grids]
    grids = [g[rng.permutation(len(g))] for g in grids]
    grid = np.concatenate(grids).astype(np.int64)
    n = len(grid)
    coord = (grid * 0.02 + rng.uniform(0, 0.02, (n, 3))).astype(np.float32)
    feat = rng.standard_normal((n, 6)).astype(np.float32)
    offset = np.cumsum([len(g) for g in grids]).astype(np.int64)
    return dict(coord=coord, grid_coord=grid, feat=feat, offset=offset)

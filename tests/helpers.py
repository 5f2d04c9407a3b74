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

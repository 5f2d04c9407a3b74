"""CPU: the curve-key header the HIP kernels compile (cdsegnet_amd/csrc/curves.h), built for the
host with gcc, must equal the numpy oracle bit for bit - validates the device bit arithmetic
without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import serialization as S
from tests.helpers import load_fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostlib():
    out = os.path.join(ROOT, "oracle", "_build", "libcurves_host.so")
    src = os.path.join(ROOT, "oracle", "c", "curves_host.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", out, src])
    return ctypes.CDLL(out)


def _enc(lib, grid, batch, depth, oid):
    grid = np.ascontiguousarray(grid, dtype=np.int64)
    out = np.zeros(len(grid), dtype=np.int64)
    b = None if batch is None else np.ascontiguousarray(batch, dtype=np.int64)
    lib.cdseg_host_encode(grid.ctypes.data_as(ctypes.c_void_p), None if b is None else b.ctypes.data_as(ctypes.c_void_p),
                          ctypes.c_long(len(grid)), ctypes.c_int(depth), ctypes.c_int(oid), out.ctypes.data_as(ctypes.c_void_p))
    return out


@pytest.mark.parametrize("depth", [1, 2, 5, 8, 9, 11, 16])
def test_random_grids(hostlib, depth):
    rng = np.random.default_rng(depth)
    g = rng.integers(0, 1 << depth, (4000, 3))
    b = rng.integers(0, 5, 4000)
    for oid, name in enumerate(S.ORDERS):
        assert np.array_equal(_enc(hostlib, g, b, depth, oid), S.encode(g, b, depth, name)), (depth, name)
        assert np.array_equal(_enc(hostlib, g, None, depth, oid), S.encode(g, None, depth, name))


@pytest.mark.parametrize("name", ["tiny64", "room1500", "batch2", "lidar5000", "rand16"])
def test_golden_codes(hostlib, name):
    fx = load_fixture(f"serialization_{name}.npz")
    for oid in range(4):
        assert np.array_equal(_enc(hostlib, fx["grid_coord"], fx["batch"], int(fx["depth"]), oid), fx["code"][oid])

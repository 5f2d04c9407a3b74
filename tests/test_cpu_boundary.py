"""CPU: the drop-in boundary - C-ABI library exports, registry contract, state_dict schema,
loud failure without a GPU.  (No compute calls: there is no GPU in the build container.)"""
import ctypes
import json
import os
import re

import pytest
import torch

from cdsegnet_amd import _lib, configs
from cdsegnet_amd.registry import MODELS, build_model
import cdsegnet_amd.models  # noqa: F401  (registers the two model names)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=_lib.VARIANTS)
def lib(request):
    """Both builds of the library: bfloat16 and IEEE half as the 16-bit type (same sources, same ABI)."""
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(_lib.LIB_PATH_F16)):
        from cdsegnet_amd.build import build_library
        build_library()
    return _lib.load(request.param)


def test_library_variant_routing_is_per_thread_and_checked():
    import threading
    from cdsegnet_amd import ops
    assert _lib.active() == "bf16"
    seen = {}
    with _lib.use("f16"):
        assert _lib.active() == "f16"
        assert ops._DT[torch.float16] == _lib.BF16 and ops._DT[torch.float32] == _lib.F32
        with pytest.raises(_lib.CdsegError):  # bfloat16 bits handed to the half build
            ops._DT[torch.bfloat16]
        t = threading.Thread(target=lambda: seen.setdefault("other", _lib.active()))
        t.start(); t.join()
    assert seen["other"] == "bf16" and _lib.active() == "bf16"
    with pytest.raises(_lib.CdsegError):
        ops._DT[torch.float16]
    with pytest.raises(ValueError):
        _lib.use("fp8")


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "cdseg.h")).read()
    declared = set(re.findall(r"\b(cdseg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cdseg_gemm_args"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cdseg_abi_version() == 1
    assert b"gfx950" in lib.cdseg_build_info()
    assert lib.cdseg_sort_ws_bytes(1000) > 0


def test_gemm_args_struct_layout_matches_header():
    # 12 pointers, one long, 14 ints, ws pointer + size, 6 LN pointers, 2 ints + float + int: any drift between include/cdseg.h and the ctypes mirror breaks every GEMM
    assert ctypes.sizeof(_lib.GemmArgs) == 12 * 8 + 8 + 14 * 4 + 16 + 6 * 8 + 16
    names = [f[0] for f in _lib.GemmArgs._fields_]
    hdr = open(os.path.join(ROOT, "include", "cdseg.h")).read()
    body = hdr[hdr.index("typedef struct cdseg_gemm_args {"):hdr.index("} cdseg_gemm_args;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    order = [m for m in re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*[;,]", body)]
    assert order == names, (order, names)


@pytest.mark.parametrize("ds", ["scannet", "scannet200", "nuscenes"])
def test_state_dict_schema_equals_reference(ds):
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", f"state_dict_schema_{ds}.json")))
    model = build_model(configs.cdsegnet_config(ds))
    mine = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(mine) == list(ref)
    assert mine == ref
    if ds == "scannet":
        assert sum(p.numel() for p in model.parameters()) == 101387354  # SURVEY.md finding 9 / paper's 101.4 M


def test_registry_contract():
    assert MODELS.get("DefaultSegmentorV2") is not None and MODELS.get("PT-v3m1") is not None
    with pytest.raises(KeyError):
        build_model(dict(type="NoSuchModel"))
    with pytest.raises(TypeError) as e:
        build_model(dict(type="PT-v3m1", no_such_kwarg=1))
    assert "PointTransformerV3" in str(e.value)  # class name prefixed like the reference's build_from_cfg
    cfg = configs.mini_config()
    cfg["backbone"]["enable_rpe"] = True
    with pytest.raises(NotImplementedError):
        build_model(cfg)


def test_load_state_dict_strict_and_engine_reset():
    from cdsegnet_amd.param_init import fill_state_dict
    model = build_model(configs.mini_config())
    sd = fill_state_dict(model.state_dict(), seed=5)
    model.load_state_dict(sd, strict=True)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cdsegnet_amd import synth
    model = build_model(configs.mini_config()).eval()
    sc = synth.room_scene(1, 300)
    inp = {k: torch.from_numpy(v) for k, v in sc.items()}
    with pytest.raises(_lib.CdsegError):
        model.inference(inp, eval=False)
    from cdsegnet_amd import ops
    with pytest.raises(_lib.CdsegError):
        ops.encode(inp["grid_coord"], None, 5, "z")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "cdsegnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "/root/reference" not in src, f


VARIANTS = [(ds, v) for ds in ("scannet", "scannet200", "nuscenes") for v in ("CDSegNet", "PTv3_CNF", "PTv3", "Baseline")]


@pytest.mark.parametrize("ds,variant", VARIANTS)
def test_model_config_equals_the_reference_config_files(ds, variant):
    """configs.model_config restates configs/<dataset>/<variant>.py; tests/golden/variant_schemas.json was produced by
    running those files themselves (oracle/make_golden.py variants) and building the reference model from them: same
    hyper-parameters, same state_dict schema (keys, order, shapes), same parameter count at FULL width."""
    import hashlib
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "variant_schemas.json")))[f"{ds}/{variant}"]
    cfg = configs.model_config(ds, variant)

    def norm(v):
        return json.loads(json.dumps(v))

    for k, v in ref["model"].items():
        assert norm(cfg.get(k)) == v, ("model", k, cfg.get(k), v)
    for k, v in ref["backbone"].items():
        assert norm(cfg["backbone"].get(k)) == v, ("backbone", k, cfg["backbone"].get(k), v)
    model = build_model(cfg)
    sd = model.state_dict()
    schema = json.dumps([[k, list(v.shape)] for k, v in sd.items()])
    assert len(sd) == ref["n_keys"]
    assert list(sd)[:3] == ref["first_keys"] and list(sd)[-3:] == ref["last_keys"]
    assert hashlib.sha256(schema.encode()).hexdigest() == ref["schema_sha256"]
    assert sum(p.numel() for p in model.parameters()) == ref["n_params"]


def test_criteria_reproduce_the_reference_loss_parts():
    """cdsegnet_amd/losses.py (MSE over the labelled points, cross entropy, multi-class Lovasz-Softmax; "EW" and "GLS"
    combinations, ref: losses/builder.py:14-52, misc.py:24-132, lovasz.py:118-265) on the predictions the REFERENCE produced
    in its recorded training step (tests/golden/train_step_mini.npz): the three parts, the GLS loss, d loss / d prediction."""
    import numpy as np
    import torch
    from cdsegnet_amd.losses import build_criteria
    from tests.helpers import load_fixture
    fx = load_fixture("train_step_mini.npz")
    cfg = [dict(type="MSELoss", loss_weight=1.0, ignore_index=-1, batch_sample_point=-1),
           dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1),
           dict(type="LovaszLoss", mode="multiclass", loss_weight=1.0, ignore_index=-1)]
    n_pred = torch.as_tensor(fx["n_pred"]).requires_grad_(True)
    c_pred = torch.as_tensor(fx["c_pred"]).requires_grad_(True)
    point = dict(n_pred=n_pred, c_pred=c_pred, c_target=torch.as_tensor(fx["noise"]), n_target=torch.as_tensor(fx["segment"]),
                 offset=torch.as_tensor(fx["offset"]), loss_mode="train")
    crit = build_criteria(cfg, loss_type="GLS", task_num=2)
    parts = np.array([float(c(point)) for c in crit.criteria])
    assert np.abs(parts - fx["loss_parts"]).max() < 1e-5
    loss = crit(point)
    assert abs(float(loss) - float(fx["loss"])) < 1e-5
    loss.backward()
    assert float((n_pred.grad - torch.as_tensor(fx["d_n_pred"])).abs().max()) < 1e-7
    assert float((c_pred.grad - torch.as_tensor(fx["d_c_pred"])).abs().max()) < 1e-7
    ew = build_criteria(cfg, loss_type="EW", task_num=2)
    assert abs(float(ew(dict(point))) - float(fx["loss_parts"].sum())) < 1e-5  # "EW": the plain sum
    point["loss_mode"] = "eval"
    assert abs(float(crit(point)) - float(fx["loss_parts"].sum())) < 1e-5      # eval mode sums whatever the loss_type
    import pytest
    with pytest.raises(NotImplementedError):
        build_criteria([dict(type="BinaryFocalLoss")])


def _shipped_plan_spec():
    """The cdseg_plan_spec of the shipped CDSegNet configs (5 n-stages of stride 2, 3 c-stages of stride 4, four curves, one
    padding key) - what Engine._native_spec builds; spelled out here so that the test does not need an engine."""
    sp = _lib.PlanSpec()
    sp.nlev = 4
    for i, c in enumerate((0, 1, 2, 3, 4)):
        sp.cum[i] = c
    sp.ncurve = 3
    for i, c in enumerate((1, 2, 3)):
        sp.curve_rows[i] = c
    sp.nslot_curve = 4
    for i, c in enumerate((-1, 0, 1, 2)):
        sp.slot_curve[i] = c
    links = [(1, 2), (2, 3), (2, 4), (3, 4)]
    sp.nlink = len(links)
    for i, (a, b) in enumerate(links):
        sp.link_a[i], sp.link_b[i] = a, b
    sp.npad = 2
    for i, (ps, fl) in enumerate(((1024, 1), (16, 0))):
        sp.pad_patch[i], sp.pad_flash[i] = ps, fl
    return sp, links


def test_native_plan_layouts_are_consistent_with_the_python_padding_arithmetic(lib):
    """Round 6: cdseg_plan_begin_layout / cdseg_plan_finish_layout are host-only - arena layouts (every item on a 256-byte
    boundary, no overlap, inside the totals) and the padding plans' statistics (K, padded length, patches, longest patch, sum
    of squared patch lengths), which the C side computes itself, against engine.Level.pad_host_py (ref: ptv3.py:188-250) on
    random ragged batches incl. elements below a patch and enable_flash = False."""
    import random
    import struct
    from cdsegnet_amd.engine import Level
    sp, links = _shipped_plan_spec()
    rnd = random.Random(7)
    for trial in range(60):
        nb = rnd.choice([1, 1, 2, 3, 8, 24])
        # pooled sizes: every level keeps at least one point per batch element and never grows
        offs = [[0]]
        for b in range(nb):
            offs[0].append(offs[0][-1] + rnd.choice([1, 7, 40, 1023, 1024, 1025, 5000, rnd.randint(1, 140000)]))
        for lvl in range(1, 5):
            row = [0]
            for b in range(nb):
                c = offs[-1][b + 1] - offs[-1][b]
                row.append(row[-1] + max(1, c // rnd.choice([1, 2, 3, 5])))
            offs.append(row)
        n = offs[0][-1]
        m = [offs[l][-1] for l in range(1, 5)]
        off_b, tot_b = (ctypes.c_long * 13)(), (ctypes.c_long * 3)()
        assert lib.cdseg_plan_begin_layout(ctypes.byref(sp), n, nb, off_b, tot_b) == 0
        sizes_b = [n, n, 3 * n, n, nb, 4 * n, 4 * (n + 1), 4 * (1 + nb) + 1, 3 * n, 1, n, n, 4 * n]
        for arena, idx in ((0, range(0, 9)), (1, range(9, 13))):
            spans = sorted((off_b[i], off_b[i] + sizes_b[i]) for i in idx)
            assert all(a % (64 if arena == 0 else 32) == 0 for a, _ in spans)
            assert all(e <= s2 for (_, e), (s2, _) in zip(spans[:-1], spans[1:])) and spans[-1][1] <= tot_b[arena]
        mh = (ctypes.c_long * 4)(*m)
        flat = [v for r in offs for v in r]
        oh = (ctypes.c_int * len(flat))(*flat)
        n_off = 3 * 4 + 2 * len(links) + 2 * 5 + 1 + 3 * 5 * 2 + 2
        off_f, info = (ctypes.c_long * n_off)(), (ctypes.c_long * (5 + 5 * 5 * 2))()
        assert lib.cdseg_plan_finish_layout(ctypes.byref(sp), n, nb, mh, oh, off_f, info) == 0
        sizes = [n] + m
        spans32, spans64 = [], []
        it = iter(off_f)
        for l in range(1, 5):
            spans32 += [(next(it), 3 * sizes[l]), (next(it), sizes[l])]
            spans64.append((next(it), 4 * sizes[l]))
        for a, b in links:
            spans32 += [(next(it), sizes[a]), (next(it), sizes[b] + 1)]
        for l in range(5):
            spans32.append((next(it), 27 * sizes[l]))
        for l in range(5):
            o = next(it)
            assert (o >= 0) == (l < 4)
            if o >= 0:
                spans64.append((o, sizes[l + 1]))
        spans32.append((next(it), 3 * sum(m)))
        q, total_slots, pads_lo = 5, 0, None
        for l in range(5):
            lv = Level(l, 10 - l, sizes[l], None, None, None, offs[l])
            for ps, fl in ((1024, True), (16, False)):
                oo, op, ot = next(it), next(it), next(it)
                K, offs_l, offs_pad, patch_start = lv.pad_host_py(ps, fl)
                lens = [b - a for a, b in zip(patch_start[:-1], patch_start[1:])]
                want = (K, offs_pad[-1], len(patch_start) - 1, max(lens), float(sum(v * v for v in lens)))
                got = tuple(info[q:q + 4]) + (struct.unpack("d", struct.pack("q", info[q + 4]))[0],)
                assert got == want, (trial, l, ps, fl, got, want)
                q += 5
                assert op == oo + nb + 1 and ot == op + nb + 1
                pads_lo = oo if pads_lo is None else pads_lo
                total_slots += 4 * offs_pad[-1]
        assert pads_lo == info[4]
        spans32.append((info[4], info[3]))
        g, w = next(it), next(it)
        spans32 += [(g, total_slots), (w, total_slots)]
        for spans, al, tot in ((spans32, 64, info[0]), (spans64, 32, info[1])):
            spans = sorted((o, o + max(1, c)) for o, c in spans)
            assert all(a % al == 0 for a, _ in spans)
            assert all(e <= s2 for (_, e), (s2, _) in zip(spans[:-1], spans[1:])) and spans[-1][1] <= tot


def test_native_plan_rejects_malformed_specs_and_sizes(lib):
    sp, _ = _shipped_plan_spec()
    off, tot = (ctypes.c_long * 13)(), (ctypes.c_long * 3)()
    assert lib.cdseg_plan_begin_layout(ctypes.byref(sp), 0, 1, off, tot) == -1          # no points
    sp.cum[2] = 1                                                                          # levels must strictly coarsen
    assert lib.cdseg_plan_begin_layout(ctypes.byref(sp), 100, 1, off, tot) == -1
    sp, _ = _shipped_plan_spec()
    sp.nlink = 3                                                                           # (3, 4) missing: level 3's parent link
    mh = (ctypes.c_long * 4)(50, 25, 12, 6)
    oh = (ctypes.c_int * 10)(0, 100, 0, 50, 0, 25, 0, 12, 0, 6)
    off_f, info = (ctypes.c_long * 64)(), (ctypes.c_long * 64)()
    assert lib.cdseg_plan_finish_layout(ctypes.byref(sp), 100, 1, mh, oh, off_f, info) == -1
    sp, _ = _shipped_plan_spec()
    oh[3] = 49                                                                             # offsets disagree with the pooled size
    assert lib.cdseg_plan_finish_layout(ctypes.byref(sp), 100, 1, mh, oh, off_f, info) == -1

#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON in the build container.

Run from the repo root:  python oracle/make_golden.py
Needs /root/reference (read-only) and therefore only works in the build container;
the produced fixtures (data only: inputs, RNG draws, expected outputs) are committed
and travel to the GPU box, the reference does not.

The reference is imported with the stand-ins of oracle/refshim (see its README for
what that pins and what it does not) and with ``torch.Tensor.cuda`` = identity
(default.py:68-72,393,400,402 hard-code ``.cuda()``).  ``enable_flash=False`` selects
the reference's own materialised-softmax CPU branch (ptv3.py:264-280).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("CDSEG_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from cdsegnet_amd import configs, synth  # noqa: E402  (data generators only)
from cdsegnet_amd.param_init import fill_state_dict  # noqa: E402


def load_reference():
    sys.path.insert(0, os.path.join(HERE, "refshim"))
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    for name, sub in (("pointcept.models", "pointcept/models"),
                      ("pointcept.models.point_prompt_training", "pointcept/models/point_prompt_training")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = m
    import pointcept  # noqa: F401
    import pointcept.models.point_prompt_training.prompt_driven_normalization as pdn
    sys.modules["pointcept.models.point_prompt_training"].PDNorm = pdn.PDNorm
    import pointcept.models.point_transformer_v3.point_transformer_v3m1_base as ptv3
    import pointcept.models.default as default
    import pointcept.models.utils.structure as structure
    import pointcept.utils.comm as comm
    return types.SimpleNamespace(ptv3=ptv3, default=default, structure=structure, comm=comm)


class DrawRecorder:
    """Record what the reference draws from the CPU generator, in order."""

    def __enter__(self):
        self.normal, self.randperm, self.randn_like = torch.normal, torch.randperm, torch.randn_like
        self.log = []
        rec = self

        def normal(*a, **k):
            r = rec.normal(*a, **k)
            rec.log.append(("normal", r.clone()))
            return r

        def randperm(*a, **k):
            r = rec.randperm(*a, **k)
            rec.log.append(("randperm", r.clone()))
            return r

        def randn_like(*a, **k):
            r = rec.randn_like(*a, **k)
            rec.log.append(("randn_like", r.clone()))
            return r

        torch.normal, torch.randperm, torch.randn_like = normal, randperm, randn_like
        return self

    def __exit__(self, *exc):
        torch.normal, torch.randperm, torch.randn_like = self.normal, self.randperm, self.randn_like


def ref_model(ref, cfg, seed=0):
    cfg = json.loads(json.dumps(cfg))  # deep copy, tuples -> lists is fine for the ctor
    cfg["backbone"]["enable_flash"] = False
    cfg["backbone"]["order"] = tuple(cfg["backbone"]["order"])
    mtype = cfg.pop("type")
    assert mtype == "DefaultSegmentorV2"
    model = ref.default.DefaultSegmentorV2(**cfg)
    sd = model.state_dict()
    new = fill_state_dict(sd, seed=seed)
    model.load_state_dict(new, strict=True)
    model.eval()
    return model, new


def save_fixture(path, **arrays):
    """np.savez_compressed, skipped when the file already holds exactly these arrays (zip bytes are not
    reproducible, array contents are): re-running the recipe leaves an unchanged fixture untouched in git."""
    if os.path.exists(path):
        with np.load(path, allow_pickle=False) as f:
            same = set(f.files) == set(arrays) and all(
                np.asarray(arrays[k]).shape == f[k].shape and np.asarray(arrays[k]).dtype == f[k].dtype and
                np.array_equal(np.asarray(arrays[k]), f[k], equal_nan=np.asarray(arrays[k]).dtype.kind == "f")
                for k in f.files)
        if same:
            print("  unchanged:", os.path.basename(path))
            return False
        print("  CHANGED:", os.path.basename(path))
    np.savez_compressed(path, **arrays)
    return True


REGEN_INPUTS = "--regen-inputs" in sys.argv  # default: inputs of an existing fixture are FROZEN (read back from it)


def frozen(tag, make):
    """Inputs of fixture `tag`: read back from the committed .npz when it exists, so that re-running the recipe
    reproduces the committed data even if cdsegnet_amd.synth has changed since (the generators are free to evolve;
    the pinned vectors are not).  ``--regen-inputs`` draws fresh inputs from synth instead."""
    path = os.path.join(OUT, f"{tag}.npz")
    if os.path.exists(path) and not REGEN_INPUTS:
        with np.load(path, allow_pickle=False) as f:
            if "coord" not in f.files:  # (a fixture that does not carry its scene: regenerate it)
                return make()
            sc = {k: f[k] for k in ("coord", "grid_coord", "feat", "offset")}
        sc["segment"] = np.zeros(len(sc["coord"]), dtype=np.int64)
        return sc
    return make()


def to_torch_input(scene):
    return dict(
        coord=torch.from_numpy(scene["coord"]),
        grid_coord=torch.from_numpy(scene["grid_coord"]),
        feat=torch.from_numpy(scene["feat"]),
        offset=torch.from_numpy(scene["offset"]),
    )


# --------------------------------------------------------------------- fixtures
def _random_cloud(seed, sizes, extent):
    """Edge case: maximum depth (16 bits per axis) and three batch elements of ragged size."""
    rng = np.random.default_rng(seed)
    scenes = []
    for n in sizes:
        g = rng.integers(0, extent, (n * 2, 3))
        g = np.unique(g, axis=0)[:n]
        g = g[rng.permutation(len(g))]
        g[0] = extent - 1
        scenes.append(dict(coord=(g * 0.01).astype(np.float32), grid_coord=g.astype(np.int64),
                           feat=rng.normal(size=(len(g), 6)).astype(np.float32),
                           segment=np.zeros(len(g), dtype=np.int64)))
    return synth.collate(scenes)


def gen_serialization(ref):
    """Point.serialization before the shuffle + padding plans + pooling structure."""
    clouds = {
        "tiny64": lambda: synth.room_scene(11, 64),
        "room1500": lambda: synth.room_scene(12, 1500),
        "batch2": lambda: synth.collate([synth.room_scene(13, 2300), synth.room_scene(14, 1200)]),
        "lidar5000": lambda: synth.lidar_scene(15, 5000),
        "rand16": lambda: _random_cloud(16, (700, 1030, 270), 65536),
        # 8 LiDAR sweeps in one batch (the nuScenes test config: batch_size_test_per_gpu = 8): depth 11 + 4 batch bits
        "lidar8": lambda: synth.collate([synth.lidar_scene(40 + i, 380 + 67 * i) for i in range(8)]),
    }
    clouds = {k: frozen(f"serialization_{k}", mk) for k, mk in clouds.items()}
    orders = ("z", "z-trans", "hilbert", "hilbert-trans")
    for name, sc in clouds.items():
        p = ref.structure.Point(dict(coord=torch.from_numpy(sc["coord"]),
                                     grid_coord=torch.from_numpy(sc["grid_coord"]),
                                     offset=torch.from_numpy(sc["offset"]),
                                     feat=torch.from_numpy(sc["feat"])))
        p.serialization(order=orders, shuffle_orders=False)
        out = dict(grid_coord=sc["grid_coord"], offset=sc["offset"], coord=sc["coord"], feat=sc["feat"],
                   batch=p.batch.numpy(), depth=np.int64(p.serialized_depth),
                   code=p.serialized_code.numpy(), order=p.serialized_order.numpy(),
                   inverse=p.serialized_inverse.numpy())
        # padding plans through the reference's own SerializedAttention helper
        for K in (4, 16, 1024):
            att = ref.ptv3.SerializedAttention(channels=16, num_heads=1, patch_size=K, enable_flash=False,
                                               upcast_attention=False, upcast_softmax=False)
            att.patch_size = K  # the non-flash ctor leaves 0 and sets it in forward
            q = ref.structure.Point(dict(offset=torch.from_numpy(sc["offset"])))
            pad, unpad, cu = att.get_padding_and_inverse(q)
            out[f"pad_K{K}"], out[f"unpad_K{K}"], out[f"cu_K{K}"] = pad.numpy(), unpad.numpy(), cu.numpy()
        # pooling structure through the reference's own SerializedPooling (no shuffle)
        for stride in (2, 4):
            bn = lambda c: torch.nn.BatchNorm1d(c, eps=1e-3)  # noqa: E731
            pool = ref.ptv3.SerializedPooling(sc["feat"].shape[1], 8, stride=stride, norm_layer=bn, act_layer=torch.nn.GELU,
                                              shuffle_orders=False)
            sdp = fill_state_dict(pool.state_dict(), seed=3)
            pool.load_state_dict(sdp)
            pool.eval()
            p.sparsify()
            with torch.no_grad():
                q = pool(p)
            out[f"pool{stride}_cluster"] = q.pooling_inverse.numpy()
            out[f"pool{stride}_grid"] = q.grid_coord.numpy()
            out[f"pool{stride}_batch"] = q.batch.numpy()
            out[f"pool{stride}_code"] = q.serialized_code.numpy()
            out[f"pool{stride}_order"] = q.serialized_order.numpy()
            out[f"pool{stride}_inverse"] = q.serialized_inverse.numpy()
            out[f"pool{stride}_feat"] = q.feat.numpy()
            out[f"pool{stride}_coord"] = q.coord.numpy()
            for k, v in sdp.items():
                out[f"pool{stride}_sd.{k}"] = v.numpy()
        save_fixture(os.path.join(OUT, f"serialization_{name}.npz"), **out)
        print("serialization", name, sc["coord"].shape, "depth", int(p.serialized_depth))
    # known answers + calc_t_emb row
    g = torch.tensor([[1, 2, 3], [7, 0, 5], [300, 2, 9]])
    from pointcept.models.utils.serialization import encode
    ka = {o: encode(g, None, 9, o).numpy() for o in orders}
    ts = 999 * torch.ones((3, 1), dtype=torch.int64)
    ka["t_emb_999_128"] = ref.comm.calc_t_emb(ts, 128).numpy()
    ka["t_emb_999_64"] = ref.comm.calc_t_emb(ts, 64).numpy()
    ts = torch.tensor([[0], [1], [500], [999]], dtype=torch.int64)
    ka["t_emb_multi_128"] = ref.comm.calc_t_emb(ts, 128).numpy()
    ka["grid"] = g.numpy()
    save_fixture(os.path.join(OUT, "known_answers.npz"), **ka)


def run_e2e(ref, cfg, scene, seed, sd_seed, tag, keep_sd, noise_level=None, capture=()):
    model, sd = ref_model(ref, cfg, seed=sd_seed)
    inp = to_torch_input(scene)
    feats = {}
    hooks = []
    for name in capture:
        mod = dict(model.named_modules())[name]

        def hook(m, i, o, name=name):
            out = o[1] if isinstance(o, tuple) else o
            feats[name] = out.feat.detach().clone().numpy()

        hooks.append(mod.register_forward_hook(hook))
    torch.manual_seed(seed)
    with DrawRecorder() as rec, torch.no_grad():
        out = model.inference(inp, eval=False, noise_level=noise_level)
    logits = out["seg_logits"].numpy()
    for h in hooks:
        h.remove()
    kinds = [k for k, _ in rec.log]
    has_noise = bool(cfg.get("dm")) and cfg.get("dm_input") == "xt"
    expect = (["randn_like"] if noise_level is not None else []) + (["normal"] if has_noise else []) + ["randperm"] * 8
    assert kinds == expect, kinds
    fx = dict(coord=scene["coord"], grid_coord=scene["grid_coord"], feat=scene["feat"], offset=scene["offset"],
              logits=logits, seed=np.int64(seed), sd_seed=np.int64(sd_seed),
              noise=([v for k, v in rec.log if k == "normal"] or [torch.zeros(0, 0)])[0].numpy(),
              perms=np.stack([v.numpy() for k, v in rec.log if k == "randperm"]),
              cfg_json=np.array(json.dumps(cfg)))
    if noise_level is not None:
        fx["feat_noise"] = rec.log[0][1].numpy()
        fx["noise_level"] = np.float64(noise_level)
    for k, v in feats.items():
        fx["trace." + k] = v
    if keep_sd:
        for k, v in sd.items():
            fx["sd." + k] = v.numpy()
    else:
        # parameters are regenerated by name (cdsegnet_amd.param_init); keep a checksum to catch drift
        fx["sd_checksum"] = np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))
        fx["sd_keys"] = np.array(list(sd.keys()))
        fx["sd_shapes"] = np.array([json.dumps(list(v.shape)) for v in sd.values()])
    save_fixture(os.path.join(OUT, f"{tag}.npz"), **fx)
    print(tag, scene["coord"].shape, "logits", logits.shape, float(np.abs(logits).mean()),
          "params", sum(v.numel() for v in sd.values()))
    return model


def gen_e2e(ref):
    cap = ("backbone._n_embedding", "backbone._n_enc.enc0", "backbone._c_enc.enc2", "backbone._n_enc.enc4",
           "backbone._tm_dec0", "backbone._n_dec.dec3", "backbone._n_dec.dec0")
    # mini model, one scene (3 patches at stage 0, padded last patch; short single patches deeper)
    run_e2e(ref, configs.mini_config(), frozen("mini_e2e_room", lambda: synth.room_scene(21, 2600)), 54421566, 1,
            "mini_e2e_room", False, capture=cap)
    # mini model, batch of two scenes (both > 1024 points so non-flash K == 1024)
    sc = frozen("mini_e2e_batch2", lambda: synth.collate([synth.room_scene(22, 1500), synth.room_scene(23, 1100)]))
    run_e2e(ref, configs.mini_config(), sc, 7, 2, "mini_e2e_batch2", False, capture=cap)
    # mini nuScenes-like (4-ch input, c target = feat)
    run_e2e(ref, configs.mini_config(num_classes=16, in_channels=4, T_dim=32),
            frozen("mini_e2e_lidar", lambda: synth.lidar_scene(24, 3000)), 99, 3, "mini_e2e_lidar", False, capture=cap)
    # mini with the reference's feature-noise knob (default.py:373-374)
    run_e2e(ref, configs.mini_config(), frozen("mini_e2e_noise", lambda: synth.room_scene(25, 1300)), 5, 4,
            "mini_e2e_noise", False, noise_level=0.1)
    # full width (101 M parameters are regenerated by name; only data is stored)
    model = run_e2e(ref, configs.cdsegnet_config("scannet"), frozen("full_e2e_8k", lambda: synth.room_scene(31, 8000)),
                    54421566, 0, "full_e2e_8k", False, capture=("backbone._n_enc.enc4", "backbone._tm_dec0"))
    schema = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_schema_scannet.json"), "w") as f:
        json.dump(schema, f, indent=0)
    for ds in ("scannet200", "nuscenes"):
        cfg = configs.cdsegnet_config(ds)
        cfg = json.loads(json.dumps(cfg))
        cfg["backbone"]["enable_flash"] = False
        cfg.pop("type")
        m = ref.default.DefaultSegmentorV2(**cfg)
        with open(os.path.join(OUT, f"state_dict_schema_{ds}.json"), "w") as f:
            json.dump({k: list(v.shape) for k, v in m.state_dict().items()}, f, indent=0)


def gen_configs(ref):
    """The BASELINE.json workload shapes and the other shipped model variants, through the reference (mini widths)."""
    # nuScenes test shape (configs/nuscenes/CDSegNet.py: batch_size_test_per_gpu = 8): EIGHT LiDAR sweeps collated
    # into one forward - grid depth 11 (0.05 m voxels over +-50 m), 4 batch bits on top of the 33 code bits, ragged
    # sweep sizes, seven of them below K = 1024 and one above (non-flash K = min n_b)
    sc = frozen("mini_e2e_lidar8", lambda: synth.collate(
        [synth.lidar_scene(50 + i, 700 + 97 * i if i else 1400) for i in range(8)]))
    run_e2e(ref, configs.mini_config(num_classes=16, in_channels=4, T_dim=32), sc, 123, 8, "mini_e2e_lidar8", False)
    # robustness shape (BASELINE config 5): coord noise sigma = 0.05 m + 50 % drop + re-voxelisation -> scattered,
    # mostly isolated voxels (few conv neighbours, different pooling ratios)
    sc = frozen("mini_e2e_robust", lambda: synth.perturb_scene(synth.room_scene(26, 5000), seed=3, sigma=0.05, drop=0.5))
    run_e2e(ref, configs.mini_config(), sc, 31, 9, "mini_e2e_robust", False)
    # configs/*/PTv3_CNF.py: CNF on the PTv3 depths (n_enc_depths (2,2,2,6,2)), linear schedule
    cfg = configs.mini_config()
    cfg["backbone"]["n_enc_depths"] = (1, 1, 2, 2, 1)
    cfg.update(beta_start=0.0001, beta_end=0.0005, noise_schedule="linear")
    run_e2e(ref, cfg, frozen("mini_cnf_room", lambda: synth.room_scene(29, 2200)), 41, 10, "mini_cnf_room", False)
    # configs/*/Baseline.py: condition=True, dm=False -> the c-branch sees the input features, t = 0, no noise draw
    cfg = configs.mini_config()
    cfg["dm"] = False
    run_e2e(ref, cfg, frozen("mini_baseline_room", lambda: synth.room_scene(30, 1900)), 43, 11, "mini_baseline_room", False)


def gen_ddim(ref):
    """Multi-step inference (MSAI mode="avg", MSFI mode="final"), default.py:278-369."""
    for tag, step, mode, seed in (("mini_ddim_avg2", 2, "avg", 17), ("mini_ddim_final1", 1, "final", 18)):
        cfg = configs.mini_config()
        model, sd = ref_model(ref, cfg, seed=6)
        scene = frozen(tag, lambda: synth.room_scene(27, 1800))
        torch.manual_seed(seed)
        with DrawRecorder() as rec, torch.no_grad():
            out = model.inference_ddim(to_torch_input(scene), T=cfg["T"], step=step, eval=False, mode=mode)
        kinds = [k for k, _ in rec.log]
        assert kinds == ["normal"] + ["randperm"] * (8 * (step + 1)), kinds
        fx = dict(coord=scene["coord"], grid_coord=scene["grid_coord"], feat=scene["feat"], offset=scene["offset"],
                  logits=out["seg_logits"].numpy(), seed=np.int64(seed), sd_seed=np.int64(6), step=np.int64(step),
                  mode=np.array(mode), noise=rec.log[0][1].numpy(),
                  perms=np.stack([v.numpy() for k, v in rec.log if k == "randperm"]),
                  cfg_json=np.array(json.dumps(cfg)),
                  sd_checksum=np.float64(sum(float(v.double().abs().sum()) for v in sd.values())),
                  sd_keys=np.array(list(sd.keys())),
                  sd_shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]))
        save_fixture(os.path.join(OUT, f"{tag}.npz"), **fx)
        print(tag, out["seg_logits"].shape, float(out["seg_logits"].abs().mean()))


def gen_ptv3(ref):
    """condition=False: plain PTv3 through the same wrapper (configs/*/PTv3.py, ptv3.py:1818-1845)."""
    cfg = configs.mini_config()
    cfg["condition"] = cfg["backbone"]["condition"] = False
    cfg["dm"] = False
    run_e2e_nocond(ref, cfg, frozen("mini_ptv3_room", lambda: synth.room_scene(28, 2100)), 21, 7, "mini_ptv3_room")


def run_e2e_nocond(ref, cfg, scene, seed, sd_seed, tag):
    model, sd = ref_model(ref, cfg, seed=sd_seed)
    torch.manual_seed(seed)
    with DrawRecorder() as rec, torch.no_grad():
        out = model.inference(to_torch_input(scene), eval=False)
    kinds = [k for k, _ in rec.log]
    assert kinds == ["randperm"] * 5, kinds
    fx = dict(coord=scene["coord"], grid_coord=scene["grid_coord"], feat=scene["feat"], offset=scene["offset"],
              logits=out["seg_logits"].numpy(), seed=np.int64(seed), sd_seed=np.int64(sd_seed),
              perms=np.stack([v.numpy() for k, v in rec.log if k == "randperm"]),
              noise=np.zeros((0, 0), dtype=np.float32), cfg_json=np.array(json.dumps(cfg)),
              sd_checksum=np.float64(sum(float(v.double().abs().sum()) for v in sd.values())),
              sd_keys=np.array(list(sd.keys())), sd_shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]))
    save_fixture(os.path.join(OUT, f"{tag}.npz"), **fx)
    print(tag, out["seg_logits"].shape, float(out["seg_logits"].abs().mean()), "params", len(sd))


def gen_gridsample(ref):
    """GridSample(mode="test") of the reference on raw (un-voxelised) clouds: datasets/transform.py:796-897."""
    import importlib.util
    # the file alone: pointcept/datasets/__init__.py pulls in loggers / dataset classes this image lacks
    spec = importlib.util.spec_from_file_location("_ref_transform", os.path.join(REF, "pointcept/datasets/transform.py"))
    T = importlib.util.module_from_spec(spec)
    sys.modules["_ref_transform"] = T  # the reference's Registry infers its scope from the defining module
    spec.loader.exec_module(T)
    for tag, seed, n, gsize in (("room", 5, 6000, 0.05), ("dense", 6, 3000, 0.2), ("neg", 7, 2000, 0.1)):
        rng = np.random.default_rng(seed)
        coord = (rng.random((n, 3)) * np.array([4.0, 3.0, 2.5])).astype(np.float32)
        if tag == "neg":
            coord -= np.array([2.0, 1.5, 0.3], dtype=np.float32)  # negative coordinates: floor, not truncation
        color = rng.random((n, 3)).astype(np.float32)
        gs = T.GridSample(grid_size=gsize, hash_type="fnv", mode="test", keys=("coord", "color"), return_grid_coord=True)
        parts = gs(dict(coord=coord.copy(), color=color.copy()))
        for p in parts:  # the transform slices every keyed array by idx_part
            assert np.array_equal(p["coord"], coord[p["index"]]) and np.array_equal(p["color"], color[p["index"]])
        idx = np.stack([p["index"] for p in parts]).astype(np.int64)
        gc = np.stack([p["grid_coord"] for p in parts]).astype(np.int64)
        save_fixture(os.path.join(OUT, f"gridsample_test_{tag}.npz"), coord=coord, grid_size=np.float64(gsize),
                            index=idx, grid_coord=gc)
        print("gridsample", tag, idx.shape)


def gen_tta(ref):
    """The reference's test pipeline up to the model: CenterShift, NormalizeColor, the 13 aug_transform lists of
    configs/scannet/CDSegNet.py:278-398, GridSample(mode="test"), CenterShift(apply_z=False) + Collect per fragment
    (datasets/defaults.py:98-132), run with the reference's own transform classes."""
    import importlib.util
    import runpy
    spec = importlib.util.spec_from_file_location("_ref_transform", os.path.join(REF, "pointcept/datasets/transform.py"))
    T = importlib.util.module_from_spec(spec)
    sys.modules["_ref_transform"] = T
    spec.loader.exec_module(T)
    cfg = runpy.run_path(os.path.join(REF, "configs/scannet/CDSegNet.py"))["data"]["test"]
    rng = np.random.default_rng(17)
    n = 1400
    coord = (rng.random((n, 3)) * np.array([3.0, 2.2, 1.6]) + np.array([1.0, -0.5, 0.2])).astype(np.float32)
    coord[:40] = coord[40:80] + rng.normal(0, 0.004, (40, 3)).astype(np.float32)  # several points per voxel
    color = rng.integers(0, 256, (n, 3)).astype(np.float32)
    normal = rng.normal(size=(n, 3)).astype(np.float32)
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    gsize = 0.05
    pre = T.Compose(cfg["transform"])
    data = pre(dict(coord=coord.copy(), color=color.copy(), normal=normal.copy()))
    vox = dict(cfg["test_cfg"]["voxelize"], grid_size=gsize)
    voxelize = T.TRANSFORMS.build(vox)
    post = T.Compose(cfg["test_cfg"]["post_transform"])
    augs = [T.Compose(a) for a in cfg["test_cfg"]["aug_transform"]]
    assert len(augs) == 13
    from copy import deepcopy
    fx = dict(coord=coord, color=color, normal=normal, grid_size=np.float64(gsize), coord0=data["coord"], color0=data["color"],
              num_aug=np.int64(len(augs)))
    for a, aug in enumerate(augs):
        d = aug(deepcopy(data))
        fx[f"aug{a}_coord"] = np.asarray(d["coord"])          # float64 after a rotation, float32 after the flip
        fx[f"aug{a}_normal"] = np.asarray(d["normal"])
        parts = voxelize(d)
        grid_full = np.full((n, 3), -1, dtype=np.int64)
        feat_full = np.zeros((n, 6), dtype=np.float32)
        sizes = []
        for k, part in enumerate(parts):
            idx = np.asarray(part["index"])
            grid_full[idx] = part["grid_coord"]
            sizes.append(len(idx))
            out = post(part)
            feat_full[idx] = out["feat"].numpy()
            if k == 0:
                fx[f"aug{a}_frag0_index"] = idx.astype(np.int64)
                fx[f"aug{a}_frag0_coord"] = out["coord"].numpy()
                assert out["offset"].tolist() == [len(idx)] and out["feat"].dtype == torch.float32
        assert (grid_full >= 0).all()
        fx[f"aug{a}_grid"] = grid_full.astype(np.int32)
        fx[f"aug{a}_feat"] = feat_full
        fx[f"aug{a}_frag_sizes"] = np.array(sizes, dtype=np.int64)
        print("tta aug", a, fx[f"aug{a}_coord"].dtype, "fragments", sizes)
    save_fixture(os.path.join(OUT, "tta_pipeline.npz"), **fx)


def gen_iou(ref):
    """intersection_and_union of the reference (utils/misc.py:38-50) on random label arrays with ignored rows."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_misc", os.path.join(REF, "pointcept/utils/misc.py"))
    M = importlib.util.module_from_spec(spec)
    sys.modules["_ref_misc"] = M
    spec.loader.exec_module(M)
    rng = np.random.default_rng(3)
    fx = {}
    for tag, n, k in (("a", 5000, 20), ("b", 777, 13), ("c", 10000, 200)):
        target = rng.integers(-1, k, size=n).astype(np.int64)  # -1 = ignore_index
        pred = np.where(rng.random(n) < 0.6, np.maximum(target, 0), rng.integers(0, k, size=n)).astype(np.int64)
        i, u, t = M.intersection_and_union(pred, target, k, -1)
        fx.update({f"{tag}_pred": pred, f"{tag}_target": target, f"{tag}_k": np.int64(k), f"{tag}_inter": i,
                   f"{tag}_union": u, f"{tag}_tgt": t})
    save_fixture(os.path.join(OUT, "iou_counts.npz"), **fx)
    print("iou fixture", {k: v.shape for k, v in fx.items() if k.endswith("inter")})


def gen_train(ref):
    """Training path, first slice: ONE Block of the reference under autograd.  The mini model runs its single-step
    forward with gradients enabled; hooks capture the residual stream in front of norm1 (x0), the qkv projection
    (retain_grad), the Block's output y and the padded patch plan of its curve; d<y, g>/d qkv comes from
    torch.autograd (what pointcept/engines/train.py:216-271's loss.backward() does for this Block)."""
    cfg = configs.mini_config()
    model, sd = ref_model(ref, cfg, seed=12)
    scene = frozen("train_block_tail", lambda: synth.room_scene(61, 2600))
    blk = model.backbone._n_enc.enc1.block0
    pre = "backbone._n_enc.enc1.block0"
    cap = {}
    hooks = [
        blk.norm1.register_forward_pre_hook(lambda m, a: cap.__setitem__("x0", a[0].feat.detach().clone())),
        blk.attn.qkv.register_forward_hook(lambda m, a, o: (o.retain_grad(), cap.__setitem__("qkv", o))[1]),
    ]

    def on_block(m, a, out):
        oi = m.attn.order_index
        pad, unpad, cu = out["pad"], out["unpad"], out["cu_seqlens_key"]
        cap["order"] = out.serialized_order[oi][pad].clone()
        cap["inverse"] = unpad[out.serialized_inverse[oi]].clone()
        cap["cu"] = cu.clone()
        cap["y"] = out.feat
    hooks.append(blk.register_forward_hook(on_block))
    torch.manual_seed(77)
    with torch.enable_grad():
        for p_ in model.parameters():
            p_.requires_grad_(True)
        model.inference(to_torch_input(scene), eval=False)
        y = cap["y"]
        g = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
        (y * g).sum().backward()
    for h in hooks:
        h.remove()
    fx = dict(coord=scene["coord"], grid_coord=scene["grid_coord"], feat=scene["feat"], offset=scene["offset"],
              x0=cap["x0"].numpy(), dy=g.numpy(), y=y.detach().numpy(), d_qkv=cap["qkv"].grad.numpy(),
              order=cap["order"].numpy().astype(np.int64), inverse=cap["inverse"].numpy().astype(np.int64),
              cu=cap["cu"].numpy().astype(np.int64), num_heads=np.int64(blk.attn.num_heads), prefix=np.array(pre))
    for k, v in sd.items():
        if k.startswith(pre + ".") and (".norm" in k or ".attn." in k or ".mlp." in k):
            fx["sd." + k] = v.numpy()
    # the reference's own parameter gradients of the tail (second slice: weight / bias / LayerNorm-affine gradients)
    for k, p_ in model.named_parameters():
        if k.startswith(pre + ".") and (".norm" in k or ".attn." in k or ".mlp." in k) and p_.grad is not None:
            fx["g." + k] = p_.grad.detach().numpy()
    save_fixture(os.path.join(OUT, "train_block_tail.npz"), **fx)
    print("train_block_tail", fx["x0"].shape, "slots", fx["order"].shape, "patches", len(fx["cu"]) - 1,
          "|d_qkv|", float(np.abs(fx["d_qkv"]).mean()))


def gen_train_step(ref):
    """The reference's TRAINING forward (default.py:424-493: per-scene timestep, q_sample, both decoders, train-mode
    BatchNorm, DropPath) with the shipped criteria (MSE + CrossEntropy + Lovasz under the GLS combination) on a batch of
    two scenes of the mini model, and loss.backward() (engines/train.py:216-271).  Recorded: every random draw in
    consumption order (timesteps, noise, order shuffles, DropPath masks per module), the loss and its parts, both
    predictions, d loss / d prediction, the L2 norm of EVERY parameter gradient and a few gradients in full."""
    from cdsegnet_amd import synth as _synth
    cfg = configs.mini_config()
    cfg["criteria"] = [dict(type="MSELoss", loss_weight=1.0, ignore_index=-1, batch_sample_point=-1),
                       dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1),
                       dict(type="LovaszLoss", mode="multiclass", loss_weight=1.0, ignore_index=-1)]
    import pointcept.models.losses  # noqa: F401  (registers the criteria)
    model, sd = ref_model(ref, cfg, seed=21)
    model.train()
    path = os.path.join(OUT, "train_step_mini.npz")
    if os.path.exists(path) and not REGEN_INPUTS:
        with np.load(path, allow_pickle=False) as f:
            scene = {k: f[k] for k in ("coord", "grid_coord", "feat", "offset", "segment")}
    else:
        scene = _synth.collate([_synth.room_scene(71, 900), _synth.room_scene(72, 700)])
        seg = np.asarray(scene["segment"]).astype(np.int64) % cfg["num_classes"]
        seg[::17] = -1  # some unlabelled points (ignore_index)
        scene = dict(coord=scene["coord"], grid_coord=scene["grid_coord"], feat=scene["feat"], offset=scene["offset"], segment=seg)
    inp = to_torch_input(scene)
    inp["segment"] = torch.from_numpy(np.asarray(scene["segment"]).astype(np.int64))
    # DropPath masks by module: the stand-in draws `bernoulli_(keep)` then divides by keep
    masks, cur = {}, [None]
    real_bernoulli = torch.Tensor.bernoulli_
    for name, mod in model.named_modules():
        if type(mod).__name__ == "DropPath" and mod.drop_prob > 0.0:
            mod.register_forward_pre_hook(lambda m, a, name=name: cur.__setitem__(0, (name, 1.0 - m.drop_prob)))

    def bernoulli_(self, *a, **k):
        r = real_bernoulli(self, *a, **k)
        name, keep = cur[0]
        masks.setdefault(name, []).append((r / keep).clone().numpy())
        return r

    real_randint = torch.randint
    ts_log = []

    def randint(*a, **k):
        r = real_randint(*a, **k)
        ts_log.append(r.clone())
        return r

    caps = {}
    torch.manual_seed(123)
    torch.Tensor.bernoulli_ = bernoulli_
    torch.randint = randint
    try:
        with DrawRecorder() as rec:
            hooks = [model.backbone._n_head.register_forward_hook(lambda m, a, o: (o.retain_grad(), caps.__setitem__("n_pred", o))[1]),
                     model.backbone._c_head.register_forward_hook(lambda m, a, o: (o.retain_grad(), caps.__setitem__("c_pred", o))[1])]
            crit = model.criteria
            parts = []
            real_call = type(crit).__call__
            for c in crit.criteria:
                c.register_forward_hook(lambda m, a, o: parts.append(float(o)))
            out = model(inp)
            loss = out["loss"]
            loss.backward()
            for h in hooks:
                h.remove()
    finally:
        torch.Tensor.bernoulli_ = real_bernoulli
        torch.randint = real_randint
    normals = [t for k, t in rec.log if k == "normal"]
    perms = [t for k, t in rec.log if k == "randperm"]
    assert len(ts_log) == 1 and len(normals) == 1 and len(perms) == 8, (len(ts_log), len(normals), len(perms))
    names = [k for k, _ in model.named_parameters()]
    gnorm = np.array([float(p_.grad.norm()) if p_.grad is not None else -1.0 for _, p_ in model.named_parameters()], dtype=np.float64)
    full = ["backbone._n_head.weight", "backbone._c_head.weight", "backbone._n_embedding.stem.conv.weight",
            "backbone._n_enc.enc2.block0.attn.qkv.weight", "backbone._n_enc.enc1.down.norm.0.weight", "backbone.fc_t1.weight",
            "backbone._tm_dec0.cross_block2.attn.kv.weight", "backbone._n_dec.dec0.up.proj.0.weight"]
    pd = dict(model.named_parameters())
    fx = dict(coord=scene["coord"], grid_coord=scene["grid_coord"], feat=scene["feat"], offset=scene["offset"],
              segment=np.asarray(scene["segment"]).astype(np.int64), sd_seed=np.int64(21),
              ts=ts_log[0].numpy(), noise=normals[0].numpy(), perms=np.stack([p_.numpy() for p_ in perms]),
              loss=np.float64(float(loss)), loss_parts=np.array(parts, dtype=np.float64),
              n_pred=caps["n_pred"].detach().numpy(), c_pred=caps["c_pred"].detach().numpy(),
              d_n_pred=caps["n_pred"].grad.numpy(), d_c_pred=caps["c_pred"].grad.numpy(),
              grad_names=np.array(names), grad_norms=gnorm)
    # one optimizer step on those gradients, grouped like pointcept/utils/optimizer.py:20-56 with the shipped settings
    # (configs/scannet/CDSegNet.py:143,152: AdamW lr 2e-3, weight decay 0.05; parameters with "block" in their name lr 2e-4)
    before = {k: p_.detach().clone() for k, p_ in model.named_parameters()}
    groups = [dict(params=[p_ for k, p_ in model.named_parameters() if "block" not in k], lr=0.002),
              dict(params=[p_ for k, p_ in model.named_parameters() if "block" in k], lr=0.0002)]
    opt = torch.optim.AdamW(groups, lr=0.002, weight_decay=0.05)
    opt.step()
    fx["step_norms"] = np.array([float((p_.detach() - before[k]).norm()) for k, p_ in model.named_parameters()], dtype=np.float64)
    for k in ("backbone._n_head.weight", "backbone._n_enc.enc2.block0.attn.qkv.weight"):
        fx["p1." + k] = pd[k].detach().numpy().copy()
    mk = sorted(masks)
    fx["mask_names"] = np.array(mk)
    fx["mask_counts"] = np.array([len(masks[k]) for k in mk], dtype=np.int64)
    for i, k in enumerate(mk):
        for j, m in enumerate(masks[k]):
            fx[f"mask.{i}.{j}"] = m.astype(np.float32)
    for k in full:
        if k in pd and pd[k].grad is not None:
            fx["g." + k] = pd[k].grad.numpy()
    save_fixture(path, **fx)
    print("train_step_mini: loss", float(loss), "parts", parts, "points", len(scene["coord"]), "DropPath modules", len(mk),
          "params with grad", int((gnorm >= 0).sum()), "of", len(gnorm))


def gen_variants(ref):
    """Full-width model variants straight from the reference's OWN config files (configs/<dataset>/<variant>.py run with
    runpy): constructor hyper-parameters the inference path reads, state_dict schema (keys + shapes, hashed) and
    parameter count of the reference model built from them -> tests/golden/variant_schemas.json.  Pins
    cdsegnet_amd.configs.model_config (our restatement of those files) - VERDICT r2 missing #5."""
    import hashlib
    import runpy
    keep_m = ("num_classes", "T", "beta_start", "beta_end", "noise_schedule", "T_dim", "dm", "dm_input", "dm_target",
              "condition", "c_in_channels", "loss_type", "task_num")
    keep_b = ("c_in_channels", "n_in_channels", "order", "c_stride", "c_enc_depths", "c_enc_channels", "c_enc_num_head",
              "c_enc_patch_size", "c_dec_depths", "c_dec_channels", "c_dec_num_head", "c_dec_patch_size", "n_stride",
              "n_enc_depths", "n_enc_channels", "n_enc_num_head", "n_enc_patch_size", "n_dec_depths", "n_dec_channels",
              "n_dec_num_head", "n_dec_patch_size", "mlp_ratio", "qkv_bias", "shuffle_orders", "pre_norm", "num_classes",
              "T_dim", "tm_feat", "condition", "skip_connection_mode", "skip_connection_scale", "skip_connection_scale_i")
    out = {}
    if "pointcept.datasets" not in sys.modules:  # the scannet200 configs import class-name constants from below it;
        m_ = types.ModuleType("pointcept.datasets")  # its __init__ pulls in every dataset reader
        m_.__path__ = [os.path.join(REF, "pointcept", "datasets")]
        sys.modules["pointcept.datasets"] = m_
    for ds in ("scannet", "scannet200", "nuscenes"):
        for variant in ("CDSegNet", "PTv3_CNF", "PTv3", "Baseline"):
            path = os.path.join(REF, "configs", ds, variant + ".py")
            cwd = os.getcwd()
            os.chdir(REF)  # _base_ includes are relative
            try:
                ns = runpy.run_path(path)
            finally:
                os.chdir(cwd)
            m = json.loads(json.dumps(ns["model"], default=str))
            b = m["backbone"]
            cfg = dict(m)
            cfg.pop("type")
            cfg["criteria"] = None
            cfg["backbone"] = dict(b, enable_flash=False, order=tuple(b["order"]))
            model = ref.default.DefaultSegmentorV2(**cfg)
            sd = model.state_dict()
            schema = json.dumps([[k, list(v.shape)] for k, v in sd.items()])
            out[f"{ds}/{variant}"] = dict(
                model={k: m.get(k) for k in keep_m if k in m}, backbone={k: b.get(k) for k in keep_b if k in b},
                n_params=int(sum(p.numel() for p in model.parameters())), n_keys=len(sd),
                schema_sha256=hashlib.sha256(schema.encode()).hexdigest(), first_keys=list(sd)[:3], last_keys=list(sd)[-3:])
            print(ds, variant, out[f"{ds}/{variant}"]["n_params"], len(sd))
    with open(os.path.join(OUT, "variant_schemas.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = load_reference()
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["ser", "e2e", "cfg", "ddim", "ptv3", "gs", "tta", "iou", "variants", "train", "trainstep"]
    if "ser" in which:
        gen_serialization(ref)
    if "e2e" in which:
        gen_e2e(ref)
    if "cfg" in which:
        gen_configs(ref)
    if "ddim" in which:
        gen_ddim(ref)
    if "ptv3" in which:
        gen_ptv3(ref)
    if "gs" in which:
        gen_gridsample(ref)
    if "tta" in which:
        gen_tta(ref)
    if "iou" in which:
        gen_iou(ref)
    if "variants" in which:
        gen_variants(ref)
    if "train" in which:
        gen_train(ref)
    if "trainstep" in which:
        gen_train_step(ref)

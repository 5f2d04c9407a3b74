"""CPU: the oracle (our restatement) against golden vectors captured from the reference's own code."""
import numpy as np
import pytest
import torch

from oracle import model as OM
from oracle import serialization as S
from tests.helpers import fixture_cfg, fixture_draws, fixture_input, fixture_state_dict, load_fixture

CLOUDS = ["tiny64", "room1500", "batch2", "lidar5000", "rand16", "lidar8"]


def test_known_answers():
    ka = load_fixture("known_answers.npz")
    g = ka["grid"]
    for o in S.ORDERS:
        assert np.array_equal(S.encode(g, None, 9, o), ka[o]), o
    # SURVEY.md 8-a5 literal values
    assert S.encode(g, None, 9, "z").tolist() == [29, 357, 67242769]
    assert S.encode(g, None, 9, "hilbert-trans").tolist() == [38, 154, 64676955]


def test_calc_t_emb():
    ka = load_fixture("known_answers.npz")
    ts = 999 * torch.ones((3, 1), dtype=torch.int64)
    assert np.array_equal(OM.calc_t_emb(ts, 128).numpy(), ka["t_emb_999_128"])
    assert np.array_equal(OM.calc_t_emb(ts, 64).numpy(), ka["t_emb_999_64"])
    ts = torch.tensor([[0], [1], [500], [999]], dtype=torch.int64)
    assert np.array_equal(OM.calc_t_emb(ts, 128).numpy(), ka["t_emb_multi_128"])


@pytest.mark.parametrize("name", CLOUDS)
def test_serialization_bit_exact(name):
    fx = load_fixture(f"serialization_{name}.npz")
    batch = S.offset2batch(fx["offset"])
    assert np.array_equal(batch, fx["batch"])
    code, order, inverse, depth = S.serialization(fx["grid_coord"], batch)
    assert depth == int(fx["depth"])
    assert np.array_equal(code, fx["code"])
    assert np.array_equal(order, fx["order"])
    assert np.array_equal(inverse, fx["inverse"])
    # codes are unique (one point per voxel) => the sort has no ties to break
    for k in range(4):
        assert len(np.unique(code[k])) == code.shape[1]


@pytest.mark.parametrize("name", CLOUDS)
@pytest.mark.parametrize("K", [4, 16, 1024])
def test_padding_plan(name, K):
    fx = load_fixture(f"serialization_{name}.npz")
    pad, unpad, cu = S.padding_plan(fx["offset"], K)
    assert np.array_equal(pad, fx[f"pad_K{K}"])
    assert np.array_equal(unpad, fx[f"unpad_K{K}"])
    assert np.array_equal(cu, fx[f"cu_K{K}"])


def test_padding_plan_survey_example():
    # n=10, K=4 -> pad=[0..9,6,7], unpad=[0..9], cu=[0,4,8,12]
    pad, unpad, cu = S.padding_plan([10], 4)
    assert pad.tolist() == list(range(10)) + [6, 7]
    assert unpad.tolist() == list(range(10))
    assert cu.tolist() == [0, 4, 8, 12]


@pytest.mark.parametrize("name", CLOUDS)
@pytest.mark.parametrize("stride", [2, 4])
def test_pooling_structure_and_values(name, stride):
    fx = load_fixture(f"serialization_{name}.npz")
    pd = {2: 1, 4: 2}[stride]
    cluster, counts, indices, idx_ptr, head, code, order, inverse = S.pooling_structure(fx["code"], pd)
    assert np.array_equal(cluster, fx[f"pool{stride}_cluster"])
    assert np.array_equal(code, fx[f"pool{stride}_code"])
    assert np.array_equal(order, fx[f"pool{stride}_order"])
    assert np.array_equal(inverse, fx[f"pool{stride}_inverse"])
    assert np.array_equal(fx["grid_coord"][head] >> pd, fx[f"pool{stride}_grid"])
    assert np.array_equal(fx["batch"][head], fx[f"pool{stride}_batch"])
    # float side of the pooling (Linear -> segment max -> BN -> GELU; segment mean of coord)
    sd = {k[len(f"pool{stride}_sd."):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(f"pool{stride}_sd.")}
    sd = {"down." + k: v for k, v in sd.items()}
    p = OM.make_point(torch.from_numpy(fx["coord"]), fx["grid_coord"], fx["offset"],
                      torch.from_numpy(fx["feat"]))
    p.code, p.order, p.inverse, p.depth = fx["code"], fx["order"], fx["inverse"], int(fx["depth"])
    q = OM.pooling(p, sd, "down", stride, None, False)
    assert np.allclose(q.feat.numpy(), fx[f"pool{stride}_feat"], atol=1e-5)
    assert np.allclose(q.coord.numpy(), fx[f"pool{stride}_coord"], atol=1e-4)


def test_subm_conv_matches_dense_conv3d():
    """spconv semantics are unpinned by the reference; pin the oracle's convention to F.conv3d."""
    rng = np.random.default_rng(0)
    for k, cin, cout in ((3, 5, 7), (5, 3, 4)):
        D = 9
        occ = rng.random((2, D, D, D)) < 0.3
        b, x, y, z = np.nonzero(occ)
        grid = np.stack([x, y, z], 1)
        feat = torch.from_numpy(rng.normal(size=(len(b), cin)).astype(np.float32))
        w = torch.from_numpy(rng.normal(size=(cout, k, k, k, cin)).astype(np.float32))
        bias = torch.from_numpy(rng.normal(size=(cout,)).astype(np.float32))
        nbr = OM.subm_neighbors(grid, b, k)
        out = OM.subm_conv3d(feat, nbr, w, bias)
        dense = torch.zeros(2, cin, D, D, D)
        dense[b, :, x, y, z] = feat
        ref = torch.nn.functional.conv3d(dense, w.permute(0, 4, 1, 2, 3).contiguous(), bias, padding=k // 2)
        ref = ref[b, :, x, y, z]
        assert torch.allclose(out, ref, atol=1e-4), (k, float((out - ref).abs().max()))


E2E = ["mini_e2e_room", "mini_e2e_batch2", "mini_e2e_lidar", "mini_e2e_noise",
       # round 2: the BASELINE workload shapes (8 collated LiDAR sweeps; noise + drop + re-voxelise) and the other
       # shipped model variants (PTv3_CNF depths / linear schedule; Baseline dm=False), all from the reference
       "mini_e2e_lidar8", "mini_e2e_robust", "mini_cnf_room", "mini_baseline_room"]


@pytest.mark.parametrize("name", E2E)
def test_e2e_mini_matches_reference(name):
    fx = load_fixture(name + ".npz")
    cfg = fixture_cfg(fx)
    sd = fixture_state_dict(fx)
    trace = {}
    nl = float(fx["noise_level"]) if "noise_level" in fx.files else None
    # the fixtures come from the reference's CPU (non-flash) branch: K = min(min_b n_b, 1024)
    dm = bool(cfg["dm"])
    logits = OM.inference(cfg["backbone"], sd, fixture_input(fx), fixture_draws(fx), T=cfg["T"],
                          noise_level=nl, run_dead=True, trace=trace, flash_semantics=False, dm=dm).numpy()
    err = np.abs(logits - fx["logits"]).max()
    assert err < 2e-4, err
    assert (logits.argmax(1) == fx["logits"].argmax(1)).mean() > 0.999
    if "trace.backbone._n_enc.enc4" in fx.files:
        assert np.abs(trace["n_enc4"].numpy() - fx["trace.backbone._n_enc.enc4"]).max() < 2e-4
        assert np.abs(trace["n_fused"].numpy() - fx["trace.backbone._tm_dec0"]).max() < 2e-4
    # the c-decoder / c-head are dead code in single-step inference
    logits2 = OM.inference(cfg["backbone"], sd, fixture_input(fx), fixture_draws(fx), T=cfg["T"],
                           noise_level=nl, run_dead=False, flash_semantics=False, dm=dm).numpy()
    assert np.array_equal(logits, logits2)
    if len(fx["offset"]) == 1:  # one batch element: flash and non-flash patching coincide
        logits3 = OM.inference(cfg["backbone"], sd, fixture_input(fx), fixture_draws(fx), T=cfg["T"],
                               noise_level=nl, flash_semantics=True, dm=dm).numpy()
        assert np.array_equal(logits, logits3)


def test_rng_replay_matches_reference_draws():
    fx = load_fixture("mini_e2e_room.npz")
    d = OM.draw_rng(int(fx["seed"]), fx["noise"].shape[0], fx["noise"].shape[1])
    assert np.array_equal(d["noise"].numpy(), fx["noise"])
    assert np.array_equal(np.stack(d["perms"]), fx["perms"])
    fx = load_fixture("mini_e2e_noise.npz")
    d = OM.draw_rng(int(fx["seed"]), fx["noise"].shape[0], fx["noise"].shape[1], noise_level_like=fx["feat"].shape)
    assert np.array_equal(d["feat_noise"].numpy(), fx["feat_noise"])
    assert np.array_equal(d["noise"].numpy(), fx["noise"])
    assert np.array_equal(np.stack(d["perms"]), fx["perms"])


def test_e2e_full_width_matches_reference():
    fx = load_fixture("full_e2e_8k.npz")
    cfg = fixture_cfg(fx)
    sd = fixture_state_dict(fx)
    logits = OM.inference(cfg["backbone"], sd, fixture_input(fx), fixture_draws(fx), T=cfg["T"]).numpy()
    err = np.abs(logits - fx["logits"]).max()
    assert err < 5e-4, err
    assert (logits.argmax(1) == fx["logits"].argmax(1)).mean() > 0.999


@pytest.mark.parametrize("name", ["mini_ddim_avg2", "mini_ddim_final1"])
def test_inference_ddim_matches_reference(name):
    """Multi-step inference (MSAI / MSFI, default.py:278-369): SURVEY.md 8f row 2."""
    fx = load_fixture(name + ".npz")
    cfg = fixture_cfg(fx)
    sd = fixture_state_dict(fx)
    draws = dict(noise=torch.from_numpy(fx["noise"]), perms=[p for p in fx["perms"]])
    out = OM.inference_ddim(cfg["backbone"], cfg, sd, fixture_input(fx), draws, step=int(fx["step"]),
                            mode=str(fx["mode"]), flash_semantics=False).numpy()
    err = np.abs(out - fx["logits"]).max()
    assert err < 5e-4, err


def test_ptv3_without_condition_matches_reference():
    """condition=False (plain PTv3 configs): SURVEY.md 8f row 4."""
    fx = load_fixture("mini_ptv3_room.npz")
    cfg = fixture_cfg(fx)
    sd = fixture_state_dict(fx)
    out = OM.inference_ptv3(cfg["backbone"], sd, fixture_input(fx), [p for p in fx["perms"]]).numpy()
    assert np.abs(out - fx["logits"]).max() < 2e-4


# ---------------------------------------------------------------- test-time pipeline
@pytest.mark.parametrize("tag", ["room", "dense", "neg"])
def test_gridsample_oracle_vs_reference(tag):
    """oracle/testtime.grid_sample_test against the reference's GridSample(mode="test") output.
    Voxel order (sorted FNV hash) and fragment count are exact; the member a fragment takes from a voxel
    depends on numpy's unstable argsort in the reference, so members are compared per voxel as sets."""
    from oracle import testtime as TT
    fx = load_fixture(f"gridsample_test_{tag}.npz")
    coord, gsize = fx["coord"], float(fx["grid_size"])
    grid, parts = TT.grid_sample_test(coord, gsize)
    ref_idx, ref_gc = fx["index"], fx["grid_coord"]
    assert len(parts) == ref_idx.shape[0]
    nvox = ref_idx.shape[1]
    voxel_of = {}
    for i, p in enumerate(parts):
        assert p.shape == (nvox,)
        assert np.array_equal(grid[p], ref_gc[i])          # same voxel in the same slot
        assert np.array_equal(grid[ref_idx[i]], ref_gc[i])  # the reference's pick lies in that voxel too
    # every voxel: the members picked over its first count_v fragments are the voxel's whole member set, in both
    ours = np.stack(parts)
    for v in range(0, nvox, max(1, nvox // 400)):
        a, b = set(ours[:, v].tolist()), set(ref_idx[:, v].tolist())
        assert a == b
    # every raw point is covered by some fragment
    assert len(np.unique(ours)) == len(coord)


def test_vote_oracle_properties():
    from oracle import testtime as TT
    rng = np.random.default_rng(0)
    n, c = 50, 7
    parts = [rng.permutation(n)[:30] for _ in range(4)]
    logits = [rng.normal(size=(30, c)).astype(np.float32) for _ in parts]
    labels, pred = TT.vote(n, c, parts, logits)
    cover = np.zeros(n)
    for p in parts:
        cover[p] += 1
    assert np.allclose(pred.sum(1), cover, atol=1e-5)  # each vote is a probability vector
    assert labels.shape == (n,)


# ---------------------------------------------------------------- evaluator
def test_iou_oracle_vs_reference_fixture():
    from oracle import testtime as TT
    fx = load_fixture("iou_counts.npz")
    for tag in ("a", "b", "c"):
        i, u, t = TT.intersection_and_union(fx[f"{tag}_pred"], fx[f"{tag}_target"], int(fx[f"{tag}_k"]), -1)
        assert np.array_equal(i, fx[f"{tag}_inter"]) and np.array_equal(u, fx[f"{tag}_union"])
        assert np.array_equal(t, fx[f"{tag}_tgt"])


def test_knn_oracle_vs_kdtree():
    from scipy.spatial import cKDTree
    from oracle import testtime as TT
    rng = np.random.default_rng(0)
    ref = rng.random((3000, 3)).astype(np.float32)
    qry = rng.random((5000, 3)).astype(np.float32)
    idx, d2 = TT.knn1_bruteforce(ref, [1200, 3000], qry, [2500, 5000])
    for (rs, re), (qs, qe) in (((0, 1200), (0, 2500)), ((1200, 3000), (2500, 5000))):
        d, j = cKDTree(ref[rs:re].astype(np.float64)).query(qry[qs:qe].astype(np.float64))
        same = (j + rs) == idx[qs:qe]
        assert same.mean() > 0.999  # float32 vs float64 near-ties only
        assert np.allclose(np.sqrt(d2[qs:qe]), d, atol=1e-5)


def test_tta_pipeline_oracle_vs_reference():
    """oracle/testtime.py's restatement of the pre-model transforms + the 13 test-time augmentations + GridSample +
    per-fragment CenterShift / Collect against tests/golden/tta_pipeline.npz, produced by the reference's own transform
    classes driven by configs/scannet/CDSegNet.py's test block (oracle/make_golden.py tta)."""
    from oracle import testtime as TT
    fx = load_fixture("tta_pipeline.npz")
    assert len(TT.SCANNET_TTA) == int(fx["num_aug"]) == 13
    coord0 = TT.center_shift(fx["coord"], apply_z=True)
    assert coord0.dtype == np.float32 and np.array_equal(coord0, fx["coord0"])
    assert np.array_equal(TT.normalize_color(fx["color"]), fx["color0"])
    res = TT.prepare_test_fragments(fx["coord"], fx["color"], fx["normal"], float(fx["grid_size"]))
    for a, r in enumerate(res):
        assert r["coord"].dtype == fx[f"aug{a}_coord"].dtype  # float64 after a rotation, float32 after the flip
        assert np.array_equal(r["coord"], fx[f"aug{a}_coord"]), a
        assert np.array_equal(r["normal"], fx[f"aug{a}_normal"]), a
        assert np.array_equal(r["grid"], fx[f"aug{a}_grid"]), a  # per-point voxel coordinates: exact
        assert [len(f["index"]) for f in r["fragments"]] == fx[f"aug{a}_frag_sizes"].tolist()
        feat = np.zeros((len(fx["coord"]), 6), dtype=np.float32)
        for f in r["fragments"]:
            feat[f["index"]] = f["feat"]
        assert np.array_equal(feat, fx[f"aug{a}_feat"]), a
        # fragment 0: same voxel set; the member chosen per voxel is platform dependent in the reference (unstable argsort)
        f0 = r["fragments"][0]
        g_ref = fx[f"aug{a}_grid"][fx[f"aug{a}_frag0_index"]]
        assert sorted(map(tuple, f0["grid_coord"])) == sorted(map(tuple, g_ref))


def test_train_oracle_matches_the_reference_autograd():
    """oracle/train.py (the checker of the training path's first slice) against the reference's own Block under
    autograd (tests/golden/train_block_tail.npz, produced by oracle/make_golden.py train)."""
    from oracle import train as OT
    fx = load_fixture("train_block_tail.npz")
    pre = str(fx["prefix"])
    sd = {k[3:]: fx[k] for k in fx.files if k.startswith("sd.")}
    y, g = OT.block_tail_qkv_grad(sd, pre, fx["x0"], fx["order"], fx["inverse"], fx["cu"], int(fx["num_heads"]), fx["dy"])
    assert float(np.abs(y.numpy() - fx["y"]).max()) < 1e-5
    assert float(np.abs(g.numpy() - fx["d_qkv"]).max()) < 1e-6
    # second slice: the oracle's parameter gradients against the ones the reference's autograd left on its parameters
    sdt = {k: torch.as_tensor(v).clone().requires_grad_(True) for k, v in sd.items()}
    x0 = torch.as_tensor(fx["x0"], dtype=torch.float32)
    yy, _ = OT.block_tail(sdt, pre, x0, fx["order"], fx["inverse"], fx["cu"], int(fx["num_heads"]))
    (yy * torch.as_tensor(fx["dy"])).sum().backward()
    checked = 0
    for k in fx.files:
        if k.startswith("g."):
            r = fx[k]
            assert float(np.abs(sdt[k[2:]].grad.numpy() - r).max()) < 1e-5 * max(1.0, float(np.abs(r).max())), k
            checked += 1
    assert checked == 12


def test_training_forward_oracle_matches_the_reference_train_step():
    """oracle/train.py::training_forward (train-mode forward of default.py:424-493 + the GLS criteria) against the
    reference's own training step on a batch of two scenes (tests/golden/train_step_mini.npz, oracle/make_golden.py
    trainstep: random draws recorded in consumption order, DropPath masks per module): loss and its three parts, both
    predictions, d loss / d prediction, and - by autograd over the restatement - the gradient of EVERY parameter (norms) and
    eight gradients in full.  The anchor for the next slices of the training path."""
    from cdsegnet_amd import configs
    from cdsegnet_amd.param_init import fill_state_dict
    from cdsegnet_amd.registry import build_model
    import cdsegnet_amd.models  # noqa: F401
    from oracle import train as OT
    fx = load_fixture("train_step_mini.npz")
    cfg = configs.mini_config()
    model = build_model(cfg)
    sd = fill_state_dict(model.state_dict(), seed=int(fx["sd_seed"]))
    names = [str(n) for n in fx["grad_names"]]
    pset = set(names)
    sdt = {k: (v.clone().float().requires_grad_(True) if k in pset else v.clone()) for k, v in sd.items()}
    masks = {str(k): [fx[f"mask.{i}.{j}"] for j in range(int(fx["mask_counts"][i]))] for i, k in enumerate(fx["mask_names"])}
    draws = dict(ts=fx["ts"], noise=fx["noise"], perms=[list(p) for p in fx["perms"]], masks=masks)
    inp = {k: fx[k] for k in ("coord", "grid_coord", "feat", "offset", "segment")}
    out = OT.training_forward(cfg, sdt, inp, draws, T=cfg["T"])
    assert abs(float(out["loss"].detach()) - float(fx["loss"])) < 1e-5
    got_parts = np.array([float(out[k].detach()) for k in ("mse", "ce", "lovasz")])
    assert np.abs(got_parts - fx["loss_parts"]).max() < 1e-5
    assert float((out["n_pred"].detach() - torch.as_tensor(fx["n_pred"])).abs().max()) < 1e-5
    assert float((out["c_pred"].detach() - torch.as_tensor(fx["c_pred"])).abs().max()) < 1e-5
    out["n_pred"].retain_grad()
    out["c_pred"].retain_grad()
    out["loss"].backward()
    assert float((out["n_pred"].grad - torch.as_tensor(fx["d_n_pred"])).abs().max()) < 1e-7
    assert float((out["c_pred"].grad - torch.as_tensor(fx["d_c_pred"])).abs().max()) < 1e-7
    gn = np.array([float(sdt[k].grad.norm()) if sdt[k].grad is not None else -1.0 for k in names])
    ref = fx["grad_norms"]
    assert (gn >= 0).all() and len(gn) == 508
    # (biases in front of a train-mode BatchNorm have an exactly-zero gradient: compare absolutely as well)
    assert (np.abs(gn - ref) <= 1e-4 * ref + 1e-6 * ref.max()).all()
    checked = 0
    for k in fx.files:
        if k.startswith("g."):
            r = fx[k]
            assert float((sdt[k[2:]].grad - torch.as_tensor(r)).abs().max()) <= 1e-5 * float(np.abs(r).max()), k
            checked += 1
    assert checked == 8
    # ... and the optimizer step that follows (AdamW, two learning-rate groups)
    worst = 0.0
    for i, k in enumerate(names):
        if ref[i] < 1e-4 * ref.max():
            continue  # (a gradient that is rounding noise - e.g. a bias in front of a train-mode BatchNorm - makes g / (|g| + eps) noise)
        lr = 0.0002 if "block" in k else 0.002
        new = OT.adamw_first_step(sd[k].float(), sdt[k].grad, lr)
        dn = float((new - sd[k].float()).norm())
        worst = max(worst, abs(dn - float(fx["step_norms"][i])) / max(float(fx["step_norms"][i]), 1e-12))
        if "p1." + k in fx.files:
            assert float((new - torch.as_tensor(fx["p1." + k])).abs().max()) < 2e-6
    assert worst < 2e-2  # (first-step AdamW moves every element by ~lr: g / (|g| + eps) amplifies gradients of ~1e-8)


def test_philox_oracle_reproduces_the_random123_known_answers():
    """oracle/philox.py (the CPU restatement of the benchmark configuration's device noise generator) against the
    known-answer vectors of the Random123 reference implementation of Philox4x32-10 (kat_vectors: zero, all-ones and the
    pi-digits counter / key)."""
    from oracle import philox as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        got = P.philox4x32_10(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))
        assert [int(v) for v in got] == list(want)
    z = P.randn(400001, 54421566, 3)
    assert z.dtype == np.float32 and z.shape == (400001,) and np.isfinite(z).all()
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01


def test_knn_bruteforce_is_consistent_with_knn1_and_sorted():
    """oracle/testtime.py::knn_bruteforce (k nearest, ascending, index order among ties, placeholders) - its first column is
    knn1_bruteforce, rows are sorted, a batch element with fewer than k points ends in placeholders."""
    from oracle import testtime as TT
    rng = np.random.default_rng(5)
    ref = rng.random((700, 3)).astype(np.float32)
    qry = rng.random((300, 3)).astype(np.float32)
    idx, d2 = TT.knn_bruteforce(6, ref, [4, 700], qry, [50, 300])
    i1, d1 = TT.knn1_bruteforce(ref, [4, 700], qry, [50, 300])
    assert np.array_equal(idx[:, 0], i1) and np.array_equal(d2[:, 0], d1)
    assert np.all(idx[:50, 4:] == -1) and np.all(d2[:50, 4:] == np.float32(1e10)) and np.all(idx[:50, :4] >= 0)
    assert np.all(idx[50:] >= 4) and np.all(np.diff(d2[50:], axis=1) >= 0)
    for r in idx[50:60]:
        assert len(set(r.tolist())) == 6

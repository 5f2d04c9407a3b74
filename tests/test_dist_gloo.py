"""CPU, world_size = 2, gloo: the N > 1 path (scene sharding, weight broadcast, counter all-reduce).
The compute inside each rank runs on the PyTorch-CPU emulation of the C-ABI ops (tests/emu_ops.py,
test infrastructure) because the product's HIP path needs a GPU; what is under test here is
cdsegnet_amd.dist and the bit-equality of the replicas after the broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cdsegnet_amd import dist as cdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cdsegnet_amd.engine as engine_mod
        import cdsegnet_amd.models  # noqa: F401
        from cdsegnet_amd import configs, synth
        from cdsegnet_amd.param_init import fill_state_dict
        from cdsegnet_amd.registry import build_model
        from oracle import model as OM
        from tests import emu_ops
        engine_mod.ops = emu_ops
        torch.set_num_threads(2)
        cfg = configs.mini_config()
        model = build_model(cfg).eval()
        if rank == 0:
            model.load_state_dict(fill_state_dict(model.state_dict(), seed=9))
        cdist.broadcast_model(model, src=0)  # rank 1 starts from its own random init
        model.precision = "fp32"
        sizes = [900, 400, 700, 300, 650]
        mine = cdist.shard_scenes(sizes)
        counts = torch.zeros(3, cfg["num_classes"], dtype=torch.int64)
        logits = {}
        for i in mine:
            sc = synth.room_scene(100 + i, sizes[i], num_classes=cfg["num_classes"])
            inp = {k: torch.as_tensor(sc[k]) for k in ("coord", "grid_coord", "feat", "offset")}
            draws = OM.draw_rng(i, sizes[i], cfg["c_in_channels"])
            out = model.inference(inp, eval=False, draws=draws)["seg_logits"]
            logits[i] = out.numpy()
            counts += cdist.confusion_counts(out.argmax(1), torch.as_tensor(sc["segment"]), cfg["num_classes"])
        cdist.reduce_counts(counts)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), counts=counts.numpy(), mine=np.array(mine),
                 **{f"logits{i}": v for i, v in logits.items()},
                 w=model.state_dict()["backbone._n_head.weight"].numpy())
    finally:
        dist.destroy_process_group()


def test_scene_sharding_is_a_partition():
    sizes = [120000, 40000, 80000, 80000, 10000, 95000, 3000]
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in cdist.shard_scenes(sizes, r, world))
        assert seen == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in cdist.shard_scenes(sizes, r, 2)) for r in range(2)]
    assert abs(loads[0] - loads[1]) <= max(sizes)


def test_two_ranks_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["w"], r1["w"]), "weights differ after the broadcast"
    assert sorted(list(r0["mine"]) + list(r1["mine"])) == [0, 1, 2, 3, 4]
    assert np.array_equal(r0["counts"], r1["counts"])  # all-reduced totals on every rank
    assert int(r0["counts"][2].sum()) == 900 + 400 + 700 + 300 + 650  # every point counted once
    # single-process reference of the same scenes with rank-0 weights
    import cdsegnet_amd.engine as engine_mod
    import cdsegnet_amd.models  # noqa: F401
    from cdsegnet_amd import configs, synth
    from cdsegnet_amd.param_init import fill_state_dict
    from cdsegnet_amd.registry import build_model
    from oracle import model as OM
    from tests import emu_ops
    old = engine_mod.ops
    engine_mod.ops = emu_ops
    try:
        cfg = configs.mini_config()
        model = build_model(cfg).eval()
        model.load_state_dict(fill_state_dict(model.state_dict(), seed=9))
        model.precision = "fp32"
        sizes = [900, 400, 700, 300, 650]
        for r in (r0, r1):
            for i in r["mine"]:
                sc = synth.room_scene(100 + int(i), sizes[int(i)], num_classes=cfg["num_classes"])
                inp = {k: torch.as_tensor(sc[k]) for k in ("coord", "grid_coord", "feat", "offset")}
                out = model.inference(inp, eval=False, draws=OM.draw_rng(int(i), sizes[int(i)], 6))["seg_logits"].numpy()
                assert np.array_equal(out, r[f"logits{int(i)}"])
    finally:
        engine_mod.ops = old


def _bcast_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cdsegnet_amd.models  # noqa: F401
        from cdsegnet_amd import configs
        from cdsegnet_amd.param_init import fill_state_dict
        from cdsegnet_amd.registry import build_model
        model = build_model(configs.mini_config()).eval()
        if rank == 0:
            model.load_state_dict(fill_state_dict(model.state_dict(), seed=3))
        ref = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
        cdist.broadcast_model(model, src=0, weight_dtype=torch.bfloat16)
        sd = model.state_dict()
        torch.save({k: v.clone() for k, v in sd.items()}, os.path.join(out_dir, f"sd{rank}.pt"))
        if rank == 0:
            torch.save(ref, os.path.join(out_dir, "ref.pt"))
    finally:
        dist.destroy_process_group()


def test_low_precision_broadcast_rounds_only_what_the_engine_casts(tmp_path):
    """a bf16 broadcast must round exactly the tensors Engine.prepare casts to the compute dtype unchanged;
    the timestep MLPs, the heads and proj_cat (scaled before its cast) travel exact, so an N-GPU run computes the
    1-GPU run's logits."""
    port = _free_port()
    mp.spawn(_bcast_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sd0, sd1, ref = (torch.load(tmp_path / f, weights_only=True) for f in ("sd0.pt", "sd1.pt", "ref.pt"))
    rounded = exact = 0
    for k, v in ref.items():
        assert torch.equal(sd0[k], sd1[k]), k  # replicas identical, src included
        if v.is_floating_point() and cdist.engine_casts(k, v):
            assert torch.equal(sd0[k], v.to(torch.bfloat16).to(v.dtype)), k
            rounded += 1
        else:
            assert torch.equal(sd0[k], v), k
            exact += 1
    assert rounded > 20 and exact > 20
    names = list(ref)
    for frag in (".t_mlp.weight", "fc_t1.weight", "fc_t2.weight", "_n_head.weight", "_c_head.weight", ".proj_cat.0.weight"):
        hit = [k for k in names if k.endswith(frag) or frag in k]
        assert hit and not any(cdist.engine_casts(k, ref[k]) for k in hit), frag
    for frag in (".attn.qkv.weight", ".cpe.0.weight", ".mlp.0.fc1.weight", ".stem.conv.weight", ".down.proj.weight"):
        hit = [k for k in names if k.endswith(frag)]
        assert hit and all(cdist.engine_casts(k, ref[k]) for k in hit), frag


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        shapes = {"head.w": (20, 64), "dec.fc2.w": (64, 256), "dec.fc2.b": (64,), "enc.conv.w": (32, 27 * 32), "emb.b": (32,),
                  "big.w": (300, 100)}
        grads = {k: torch.randn(*sh, generator=g) for k, sh in shapes.items()}
        b = cdist.GradBucketer(bucket_bytes=64 * 1024)  # small buckets: several flushes, one tensor larger than a bucket
        for k in shapes:  # the order a backward would hand them over
            b.add(k, grads[k])
        out = b.finish()
        torch.save({"out": {k: v.clone() for k, v in out.items()}, "mine": grads, "buckets": b.buckets_reduced},
                   os.path.join(out_dir, f"grads{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_grad_bucketer_two_ranks_gloo(tmp_path):
    """Bucketed gradient all-reduce of the training path: both ranks end with the mean, tensor by tensor."""
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"grads{r}.pt")) for r in (0, 1))
    assert r0["buckets"] >= 3
    for k in r0["mine"]:
        mean = (r0["mine"][k] + r1["mine"][k]) / 2
        assert r0["out"][k].shape == mean.shape
        assert torch.allclose(r0["out"][k], mean, atol=1e-6) and torch.equal(r0["out"][k], r1["out"][k])
    one = cdist.GradBucketer()  # no process group: pass-through
    one.add("w", torch.ones(3, 3))
    assert torch.equal(one.finish()["w"], torch.ones(3, 3))


def _gather_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [900, 400, 700, 300, 650]
        mine = cdist.shard_scenes(sizes)
        if rank == 1:
            mine = []  # a rank without scenes takes part in the collectives all the same
        items = [(i, (torch.arange(sizes[i]) * (i + 3) % 20 - 1).to(torch.int64)) for i in mine]
        got = cdist.gather_predictions(items)
        np.savez(os.path.join(out_dir, f"g{rank}.npz"), ids=np.array(sorted(got)), **{f"s{i}": v.numpy() for i, v in got.items()})
    finally:
        dist.destroy_process_group()


def test_prediction_gather_three_ranks_gloo(tmp_path):
    """cdist.gather_predictions: every rank ends up with every scene's labels (int16), ragged scene sizes, an idle rank."""
    port = _free_port()
    mp.spawn(_gather_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    sizes = [900, 400, 700, 300, 650]
    on_idle = set(cdist.shard_scenes(sizes, 1, 3))
    want = sorted(set(range(5)) - on_idle)
    for r in range(3):
        g = np.load(tmp_path / f"g{r}.npz")
        assert list(g["ids"]) == want
        for i in want:
            assert g[f"s{i}"].dtype == np.int16
            assert np.array_equal(g[f"s{i}"], (np.arange(sizes[i]) * (i + 3) % 20 - 1).astype(np.int16))
    single = cdist.gather_predictions([(7, torch.tensor([1, -1, 19]))])
    assert list(single) == [7] and single[7].dtype == torch.int16


def _train_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cdsegnet_amd.engine as engine_mod
        import cdsegnet_amd.models  # noqa: F401
        import cdsegnet_amd.train_graph as tg
        from cdsegnet_amd import configs, synth
        from cdsegnet_amd.param_init import fill_state_dict
        from cdsegnet_amd.registry import build_model
        from tests import emu_ops
        engine_mod.ops = emu_ops
        tg.ops = emu_ops
        torch.set_num_threads(2)
        cfg = configs.mini_config()
        cfg["backbone"]["enable_flash"] = False
        cfg["criteria"] = [dict(type="MSELoss", loss_weight=1.0, ignore_index=-1, batch_sample_point=-1),
                           dict(type="CrossEntropyLoss", loss_weight=1.0, ignore_index=-1),
                           dict(type="LovaszLoss", mode="multiclass", loss_weight=1.0, ignore_index=-1)]
        model = build_model(cfg).train()
        if rank == 0:
            model.load_state_dict(fill_state_dict(model.state_dict(), seed=3))
        cdist.broadcast_model(model, src=0)

        def batch(i):  # rank i's scene (seeded: any rank can rebuild any batch) and its recorded draws
            sc = synth.room_scene(300 + i, 500 + 150 * i, num_classes=cfg["num_classes"])
            inp = {k: torch.as_tensor(sc[k]) for k in ("coord", "grid_coord", "feat", "offset", "segment")}
            g = torch.Generator().manual_seed(40 + i)
            n = inp["feat"].shape[0]
            draws = dict(ts=torch.randint(0, cfg["T"], (1, 1), generator=g), noise=torch.randn(n, cfg["c_in_channels"], generator=g),
                         perms=[torch.randperm(4, generator=g).tolist() for _ in range(8)], masks={})  # masks {}: no stochastic depth
            return inp, draws

        sync = cdist.GradSync(model, bucket_bytes=256 * 1024)  # small buckets: several all-reduces per backward
        inp, draws = batch(rank)
        loss = model(inp, draws=draws)["loss"]
        loss.backward()
        sync.finish()
        got = {k: p.grad.clone() for k, p in model.named_parameters()}
        nb = sync.buckets_reduced
        sync.remove()
        ref = None
        if rank == 0:  # single-process reference: the mean of the two ranks' gradients
            ref = {}
            for i in range(world):
                model.zero_grad()
                inp, draws = batch(i)
                model(inp, draws=draws)["loss"].backward()
                for k, p in model.named_parameters():
                    ref[k] = ref.get(k, 0) + p.grad / world
        torch.save(dict(got=got, ref=ref, loss=float(loss.detach()), buckets=nb), os.path.join(out_dir, f"train{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_data_parallel_training_step_two_ranks_gloo(tmp_path):
    """Two ranks, one scene each: the training forward + loss.backward() of cdsegnet_amd/train_graph.py with GradSync's
    bucketed all-reduce (hooked on the parameters) leaves the SAME averaged gradients on both ranks, equal to the mean of the
    two scenes' gradients computed in one process (ref: DDP in engines/defaults.py:38 + engines/train.py:216-271)."""
    port = _free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "train0.pt"), torch.load(tmp_path / "train1.pt")
    assert r0["buckets"] == r1["buckets"] and r0["buckets"] >= 2
    assert abs(r0["loss"] - r1["loss"]) > 1e-6  # different scenes
    assert set(r0["got"]) == set(r1["got"]) and len(r0["got"]) == 508
    for k in r0["got"]:
        assert torch.equal(r0["got"][k], r1["got"][k]), k
        r = r0["ref"][k]
        assert float((r0["got"][k] - r).abs().max()) <= 1e-5 * max(1e-3, float(r.abs().max())), k


def _order_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(8, 8, bias=False), torch.nn.Linear(8, 8, bias=False))
        sync = cdist.GradSync(model, bucket_bytes=1 << 20)
        res = {}
        # step 1: both ranks hand their gradients over in the same order; step 2: rank 1 in the opposite order (two
        # parameters of equal size, so the all-reduces still pair up - only the digest can notice)
        for step, flip in ((1, False), (2, rank == 1)):
            model.zero_grad()
            for p in model.parameters():
                p.grad = torch.full_like(p, float(rank + 1))
            params = list(model.named_parameters())
            for (n, p) in (params[::-1] if flip else params):
                sync._hook(n)(p)
            try:
                sync.finish()
                res[step] = float(model[0].weight.grad[0, 0])
            except RuntimeError as e:
                res[step] = str(e)
        torch.save(res, os.path.join(out_dir, f"order{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_grad_sync_notices_ranks_that_disagree_on_the_gradient_order(tmp_path):
    """The order digest rides behind the last gradient bucket (no collective of its own): agreeing ranks get the
    mean, ranks whose hand-over order differs get a RuntimeError on every rank."""
    port = _free_port()
    mp.spawn(_order_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(str(tmp_path), f"order{r}.pt"))
        assert res[1] == 1.5
        assert isinstance(res[2], str) and "different orders" in res[2]


def test_rank_host_plan_on_a_mocked_8_gpu_two_socket_node(tmp_path):
    """what each of 8 ranks does to the HOST (the only thing they share): CPUs of its GPU's NUMA node,
    split among the ranks of that node; a cap on torch's intra-op threads; blocking host reads.  Mocked topology: GPUs 0-3 on
    node 0, 4-7 on node 1, 2 x 64 cores with SMT siblings numbered +128 (the GPU boxes' 256 hardware threads)."""
    node0 = list(range(0, 64)) + list(range(128, 192))
    node1 = list(range(64, 128)) + list(range(192, 256))
    gpu_numa, numa_cpus, all_cpus = [0, 0, 0, 0, 1, 1, 1, 1], {0: node0, 1: node1}, list(range(256))
    plans = [cdist.rank_host_plan(r, 8, gpu_numa, numa_cpus, all_cpus) for r in range(8)]
    seen = set()
    for r, p in enumerate(plans):
        assert p["numa_node"] == gpu_numa[r] and len(p["cpus"]) == 32 and p["threads"] == 8 and p["blocking_sync"]
        assert set(p["cpus"]) <= set(numa_cpus[gpu_numa[r]]) and not (set(p["cpus"]) & seen)
        seen |= set(p["cpus"])
    assert seen == set(range(256))  # every hardware thread belongs to exactly one rank
    # a container that may only use 8 CPUs of node 0: nobody gets an empty set, node-1 ranks fall back to the allowed CPUs
    small = [cdist.rank_host_plan(r, 8, gpu_numa, numa_cpus, list(range(8))) for r in range(8)]
    assert all(p["cpus"] and set(p["cpus"]) <= set(range(8)) and p["threads"] >= 1 for p in small)
    assert sorted(c for p in small[:4] for c in p["cpus"]) == list(range(8))
    # no NUMA information (sysfs says -1): an even split of everything
    flat = [cdist.rank_host_plan(r, 4, [-1] * 4, {}, list(range(64))) for r in range(4)]
    assert [p["cpus"] for p in flat] == [list(range(16 * r, 16 * r + 16)) for r in range(4)]
    # one rank: the whole local node, nothing to block for
    one = cdist.rank_host_plan(0, 1, [1], numa_cpus, all_cpus, max_threads=16)
    assert one["cpus"] == sorted(node1) and one["threads"] == 16 and not one["blocking_sync"]
    # the sysfs reader on a fake tree
    sysfs = tmp_path / "sys"
    for bus, node in (("0000:05:00.0", 0), ("0000:c5:00.0", 1)):
        d = sysfs / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    for node, text in ((0, "0-63,128-191"), (1, "64-127,192-255")):
        d = sysfs / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(text + "\n")
    g, nc, _ = cdist.read_host_topology(["0000:05:00.0", "0000:C5:00.0", "0000:ff:00.0"], sysfs=str(sysfs))
    assert g == [0, 1, -1] and nc == {0: node0, 1: node1}
    # applying a plan in this process (no GPU here): affinity + thread cap, restored afterwards
    import os as _os
    keep_aff, keep_thr = _os.sched_getaffinity(0), torch.get_num_threads()
    try:
        mine = sorted(keep_aff)
        p = cdist.rank_host_plan(1, 2, [-1, -1], {}, mine, max_threads=2)
        applied = cdist.apply_rank_host_plan(p, set_device_flags=False)
        assert applied["affinity"] and _os.sched_getaffinity(0) == set(p["cpus"]) and torch.get_num_threads() == p["threads"]
    finally:
        _os.sched_setaffinity(0, keep_aff)
        torch.set_num_threads(keep_thr)

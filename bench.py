#!/usr/bin/env python3
"""bench.py - CDSegNet single-step inference throughput on MI355X (points/s/node).

    python bench.py --gpus N --steps K --warmup W

N > 1 with no torch.distributed environment: bench.py launches its own N ranks (re-executes itself under
torch.distributed.run on 127.0.0.1, like the reference's self-spawning launcher, pointcept/engines/launch.py:35-135);
under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it reads RANK / LOCAL_RANK / WORLD_SIZE.

A step = one SSI pass (DefaultSegmentorV2.inference: PTv3 dual backbone + cross-attention fusion) over one batch of
--scenes-per-forward x --lanes (8 x 3 = 24) DISTINCT synthetic ScanNet-shaped scenes per GPU (BASELINE.json configs[1]:
~120k voxels each - sizes 103k..137k, mean 120k - 6-ch features, 20 classes, IEEE-half trunk + fp32 heads by default), inputs already resident in HBM.
Scenes are independent units: every rank runs its own scenes, no data-path collective ("scaling": "weak").  Every lane
(HIP stream) gets one collated forward of 8 scenes (the reference's collate_fn batching), three forwards are in flight.
`value` counts all points of all scenes; `single_scene_latency_ms` is the one-scene-at-a-time (bs = 1) latency, the
median of 9 synchronised calls.
RCCL is used once to broadcast the weights from rank 0 and for the final timing / counter reductions.

Prints ONE JSON line on rank 0.  `roofline` (window attention, the north-star kernel) and `roofline_conv` (all k = 3 sparse
convs of the forward) are timed with HIP events around every launch ON THE LAUNCH STREAM, in a pass
right after the timed region that replays the timed configuration's own forward (the same 8 collated scenes) with
nothing else on the GPU; `cpu_baseline` is the CPU oracle on the host cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the first HIP call: one hardware queue per lane (see cdsegnet_amd)
# dmabuf IPC: RCCL needs it on this driver - set here (before torch is imported), so a rank started by an EXTERNAL
# `python -m torch.distributed.run ... bench.py` has it too, not only the ranks of self_launch()
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "bf16+head": 2500.0, "fp16": 2500.0, "fp16+head": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
CPU_THREADS = 16  # fastest of {8, 16, 32} torch threads on the GPU box host at the baseline's own 120 k-voxel scene (tools/cpu_sweep.py, profiles/r05_cpu_sweep.txt; r02: 24 k voxels)


def dtype_name(precision):
    """The arithmetic type of the path for the bench line's `dtype`."""
    return "f32" if precision == "fp32" else ("f16" if precision.startswith("fp16") else "bf16")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=120000, help="mean voxels per scene")
    ap.add_argument("--dataset", default="scannet", choices=["scannet", "scannet200", "nuscenes"])
    ap.add_argument("--robust", action="store_true",
                    help="BASELINE config 5: every scene gets Gaussian coord noise sigma = 0.05 m + 50 %% random drop and is "
                         "re-voxelised (~half the points, scattered voxels)")
    ap.add_argument("--precision", default="fp16+head", choices=["fp16+head", "fp16", "bf16+head", "bf16", "fp32"],
                    help="<16-bit type>[+head]: 16-bit MFMA operands and activations (fp16 = IEEE half, the reference's own "
                         "attention dtype, 11-bit mantissa; bf16 = bfloat16, 8-bit), fp32 accumulation and residual stream; "
                         "+head: the logit head GEMM in exact fp32 on the fp32 stream (free, profiles/r03_bf16_budget.txt); "
                         "fp32: the 1e-3 parity mode")
    ap.add_argument("--protocol", default="throughput", choices=["throughput", "paper"],
                    help="paper: the reference's timing protocol (tools/test_time.py:36-37,79 + configs/scannet/"
                         "CDSegNet_time.py): 312 distinct scenes, one at a time (bs = 1), no TTA, wall clock")
    ap.add_argument("--cpu-baseline", dest="cpu_baseline", action="store_true", default=True)
    ap.add_argument("--no-cpu-baseline", dest="cpu_baseline", action="store_false")
    ap.add_argument("--cpu-points", type=int, default=120000)
    ap.add_argument("--cpu-threads", type=int, default=CPU_THREADS)
    ap.add_argument("--no-kernel-timer", action="store_true", help="skip the roofline pass after the timed region")
    ap.add_argument("--no-agreement", action="store_true", help="skip the 16-bit-vs-fp32 agreement leg")
    ap.add_argument("--no-paper-pass", action="store_true",
                    help="skip the separate `--protocol paper` process whose result the default line carries as paper_protocol.own_process")
    ap.add_argument("--scenes-per-forward", type=int, default=24,
                    help="scenes collated into one forward (the reference's collate_fn batching); one step = "
                         "scenes-per-forward x lanes scenes (one batch per lane)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="independent forwards in flight per GPU (HIP streams); 1 = strictly one forward at a time")
    ap.add_argument("--serial", action="store_true",
                    help="profiling aid: no side-stream fork inside a forward (with --lanes 1 no two kernels ever overlap, so a "
                         "rocprofv3 --kernel-trace of the run shows each kernel's own duration)")
    ap.add_argument("--shard", type=int, default=0, metavar="S",
                    help="strong-scaling mode (BASELINE config 4's shape: --dataset nuscenes --points 40000 --shard 64): ONE "
                         "global list of S mixed-size scenes, LPT-sharded over the ranks (cdsegnet_amd.dist.shard_scenes), "
                         "every rank runs its share through inference_many (--scenes-per-forward collated per forward), the "
                         "per-class counters are all-reduced and the mIoU printed; a step = all S scenes")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group plumbing only (no model, gloo when there is no GPU): what the CPU test runs")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """--gpus N without a torch.distributed environment: become the launcher of N ranks on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    return subprocess.call(cmd, env=env)


def make_scenes(args, rank, count):
    """`count` distinct scenes of this rank: different seeds AND sizes (mean = --points), so ragged patches, LPT
    sharding and the allocator see what a real scene list gives them."""
    from cdsegnet_amd import synth
    scenes = []
    for i in range(count):
        seed = 1000 * rank + i
        if args.dataset == "nuscenes":
            n = int(round(args.points * (1.0 + 0.25 * ((i + 0.5) / count - 0.5))))
            sc = synth.lidar_scene(seed, n)
        else:
            n = int(round(args.points * (1.0 + 0.3 * ((i + 0.5) / count - 0.5))))  # 0.85 .. 1.15 x, mean 1.0
            sc = synth.room_scene(seed, n)
        if args.robust:
            sc = synth.perturb_scene(sc, seed=seed, sigma=0.05, drop=0.5, voxel=0.05 if args.dataset == "nuscenes" else 0.02)
        scenes.append(sc)
    return scenes


def shard_sizes(args):
    """Target sizes of the global scene list of --shard: deterministic, mixed (0.6 .. 1.4 x --points)."""
    return [int(round(args.points * (0.6 + 0.8 * ((i * 37) % args.shard) / max(1, args.shard - 1)))) for i in range(args.shard)]


def cpu_baseline(cfg, sd, points, dataset, threads):
    """The CPU oracle (our PyTorch-CPU fp32 restatement of the reference path) on ONE scene of the bench generator:
    one warm-up, median of three, on `threads` torch threads (swept once: tools/cpu_sweep.py)."""
    from cdsegnet_amd import synth
    from oracle import model as OM
    sc = synth.lidar_scene(100, points) if dataset == "nuscenes" else synth.room_scene(100, points)
    n = len(sc["coord"])
    inp = {k: sc[k] for k in ("coord", "grid_coord", "feat", "offset")}
    draws = OM.draw_rng(1, n, cfg["c_in_channels"])
    keep = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        times = []
        ref = None
        for i in range(4):
            t0 = time.time()
            ref = OM.inference(cfg["backbone"], sd, inp, draws, T=cfg["T"])
            if i:
                times.append(time.time() - t0)
    finally:
        torch.set_num_threads(keep)
    med = float(np.median(times))
    out = dict(value=n / med, unit="points/s", cores=threads, kind="port",
               sample=f"1 scene x {n} points (bench generator and model, fp32, PyTorch-CPU oracle), 1 warm-up + median of 3: "
                      f"{med:.1f} s (runs {', '.join(f'{t:.1f}' for t in times)} s), {threads} of {os.cpu_count()} host threads")
    return out, (inp, draws, ref)


def parity_vs_cpu(model, dev, cpu_case, precisions):
    """The HIP path on the cpu_baseline leg's own scene and draws against the CPU oracle's logits of that run (the oracle
    as the checker, never as the thing measured): max |logit difference| and arg-max agreement per precision, at the
    full 120k-point size - north_star's float bound (1e-3) is stated for the fp32 mode."""
    inp, draws, ref = cpu_case
    ref = ref.to(dev)
    d = {k: torch.as_tensor(v).to(dev) for k, v in inp.items()}
    keep_p, keep_n = model.precision, model.noise_source
    out = {}
    try:
        model.noise_source = "torch_cpu"
        for pr in precisions:
            model.precision = pr
            got = model.inference(dict(d), eval=False, draws=dict(draws))["seg_logits"]
            out[pr] = dict(max_abs_logit_err_vs_cpu_oracle=float((got - ref).abs().max()),
                           argmax_agreement_vs_cpu_oracle=float((got.argmax(1) == ref.argmax(1)).float().mean()))
    finally:
        model.precision, model.noise_source = keep_p, keep_n
    return out


def agreement_leg(model, scene_dict, n, cfg, precision):
    """One bench scene through `precision` and through the exact-fp32 HIP path on the same draws: arg-max agreement and
    logit differences (random-init weights give near-ties: the margins say how far apart the flipped classes were)."""
    keep_p, keep_n = model.precision, model.noise_source
    gen = torch.Generator().manual_seed(54421566)
    draws = dict(noise=torch.normal(0, 1, size=(n, cfg["c_in_channels"]), dtype=torch.float32, generator=gen),
                 perms=[torch.randperm(4, generator=gen).tolist() for _ in range(8)])
    try:
        model.noise_source = "torch_cpu"
        model.precision = precision
        a = model.inference(dict(scene_dict), eval=False, draws=dict(draws))["seg_logits"].clone()
        model.precision = "fp32"
        b = model.inference(dict(scene_dict), eval=False, draws=dict(draws))["seg_logits"]
    finally:
        model.precision, model.noise_source = keep_p, keep_n
    top2 = b.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]  # fp32 top-1 / top-2 margin of every point
    flipped = a.argmax(1) != b.argmax(1)
    return dict(points=n, precision=precision,
                argmax_agreement=float((~flipped).float().mean()),
                max_abs_logit_diff=float((a - b).abs().max()), rms_logit_diff=float((a - b).pow(2).mean().sqrt()),
                mean_abs_logit=float(b.abs().mean()),
                fp32_margin_median=float(margin.median()),
                points_with_margin_below_0p01=float((margin < 0.01).float().mean()),
                largest_margin_that_flipped=float(margin[flipped].max()) if bool(flipped.any()) else 0.0,
                reference="exact-fp32 HIP path (within 5e-6 of the reference's CPU logits, tests/)")


def attention_ceiling(achieved_tflops):
    """ONE ceiling for the attention kernel, from the issue model of its key loop: a 32 x 32 score tile
    (65 536 algorithmic FLOP) is 16 v_exp_f32 + 8 v_cvt_pk_bf16_f32 next to 3 MFMAs, and the VALU / transcendental side is the
    longer one - tools/ubench/pipes.hip runs exactly that mix with no dependences and no memory, 4 waves per SIMD on every
    SIMD: `cycles_per_tile_simd` (tracked: profiles/r03_ubench_pipes.txt).  At the clock the KERNEL holds (GRBM_GUI_ACTIVE /
    duration of an offline rocprofv3 --pmc pass, profiles/r03_attention_clock.json) that is the ceiling in TFLOP/s;
    `frac_of_ceiling` = achieved / ceiling.  (Context, not a second fraction: run flat out, the mix itself pulls the chip down
    to ~1.5 GHz - `mix_alone_ns_per_tile_simd` - so in wall time the ceiling is lower still.)"""
    import re
    try:
        txt = open(os.path.join(ROOT, "profiles", "r03_ubench_pipes.txt")).read()
        m = re.search(r"round-3 tile\)\s+W=1:.*?W=4:\s+([0-9.]+) cyc/iter/SIMD\s+([0-9.]+) ns \(([0-9.]+) GHz\)", txt)
        cyc, mix_ns, mix_ghz = float(m.group(1)), float(m.group(2)), float(m.group(3))
        clock = float(json.load(open(os.path.join(ROOT, "profiles", "r03_attention_clock.json")))["ghz"])
    except Exception:  # noqa: BLE001 - both files are part of the repository; without them the line carries no ceiling
        return None
    tflops = 65536.0 * 1024.0 * clock * 1e9 / cyc / 1e12  # 1024 SIMDs
    return {"tflops": tflops, "frac_of_peak": tflops / PEAK_TFLOPS["bf16"], "frac_of_ceiling": achieved_tflops / tflops,
            "model": "16 v_exp_f32 + 8 v_cvt_pk + 3 MFMA per 32x32 tile: VALU / transcendental issue bound",
            "cycles_per_tile_simd": cyc, "kernel_clock_ghz": clock, "mix_alone_ns_per_tile_simd": mix_ns,
            "mix_alone_clock_ghz": mix_ghz,
            "sources": "profiles/r03_ubench_pipes.txt ('3 mfma + 16 exp + 8 cvt_pk', W = 4), profiles/r03_attention_clock.json"}


def paper_protocol(args, model, cfg, dev, rank, world, dist):
    """--protocol paper: 312 DISTINCT synthetic scenes (the ScanNet val split's count), one at a time, wall clock -
    the reference's tools/test_time.py:36-37,79 with configs/scannet/CDSegNet_time.py (bs = 1, no TTA)."""
    from cdsegnet_amd import synth
    n_scenes = 312
    mine = list(range(rank, n_scenes, world))
    dicts, pts = [], 0
    for i in mine:
        n = int(round(args.points * (1.0 + 0.3 * ((i % 24 + 0.5) / 24 - 0.5))))
        sc = synth.lidar_scene(5000 + i, n) if args.dataset == "nuscenes" else synth.room_scene(5000 + i, n)
        d = {k: torch.as_tensor(sc[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset")}
        pts += len(sc["coord"])
        dicts.append(d)
    torch.manual_seed(54421566 + rank)
    for d in dicts[:3]:
        model.inference(dict(d), eval=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for d in dicts:  # the reference's input dict (no offset_host hint): every host sync of the call is inside
        model.inference(dict(d), eval=False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    tp = torch.tensor([pts], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tp, op=dist.ReduceOp.SUM)
    if rank == 0:
        sec = float(el.item())
        print(json.dumps({
            "metric": "seconds for the 312-scene val split, 1-step inference, bs = 1 (reference protocol tools/test_time.py)",
            "value": sec, "unit": "s", "n_gpus": world, "steps": n_scenes, "warmup": 3, "ms_per_step": 1e3 * sec * world / n_scenes,
            "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": dtype_name(args.precision),
            "data": "synthetic",
            "config": {"workload": f"312 distinct synthetic ScanNet-shaped scenes, one at a time, no TTA, {int(tp.item()) / n_scenes:.0f} voxels mean",
                       "precision": args.precision, "reference_figure": "56 s on an RTX 3090 (BASELINE.md; real scans, other hardware)"},
            "points_per_s": int(tp.item()) / sec}))
    if world > 1:
        dist.destroy_process_group()


def shard_mode(args, model, cfg, dev, rank, world, dist):
    """--shard S: strong scaling over ONE global list of S scenes (see parse())."""
    from cdsegnet_amd import dist as cdist, synth
    sizes = shard_sizes(args)
    mine = cdist.shard_scenes(sizes, rank, world)
    scenes = []
    for i in mine:
        sc = synth.lidar_scene(9000 + i, sizes[i]) if args.dataset == "nuscenes" else synth.room_scene(9000 + i, sizes[i])
        scenes.append(sc)
    dicts = []
    for sc in scenes:
        d = {k: torch.as_tensor(sc[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset")}
        d["offset_host"] = [int(v) for v in sc["offset"]]
        dicts.append(d)
    my_pts = int(sum(len(sc["coord"]) for sc in scenes))
    torch.manual_seed(54421566 + rank)

    def step():
        return model.inference_many([dict(d) for d in dicts], lanes=args.lanes, batch=args.scenes_per_forward) if dicts else []

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = None
    for _ in range(args.steps):
        outs = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    counts = torch.zeros(3, cfg["num_classes"], dtype=torch.int64, device=dev)
    for sc, o in zip(scenes, outs or []):
        counts += cdist.confusion_counts(o["seg_logits"].argmax(1), torch.as_tensor(sc["segment"]).to(dev), cfg["num_classes"])
    cdist.reduce_counts(counts)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    pts = torch.tensor([my_pts], dtype=torch.int64, device=dev)
    per_rank = torch.zeros(world, dtype=torch.int64, device=dev)
    per_rank[rank] = my_pts
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(pts, op=dist.ReduceOp.SUM)
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
    if rank == 0:
        el = float(tmax.item())
        m = cdist.metrics(counts)
        print(json.dumps({
            "metric": "points/sec/node (one global scene list sharded over the GPUs, 1-step)", "value": int(pts.item()) * args.steps / el,
            "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype_name(args.precision),
            "data": "synthetic",
            "config": {"workload": f"{args.shard} {args.dataset}-shaped scenes ({min(sizes)}..{max(sizes)} voxels), LPT-sharded over "
                                   f"{world} GPU(s), {args.scenes_per_forward} collated per forward, {args.lanes} forwards in flight",
                       "precision": args.precision, "points_per_rank": per_rank.tolist(), "host_hints": ["offset_host"]},
            "scenes_per_s": args.shard * args.steps / el,
            "eval_counters": {"mIoU_random_init": m["mIoU"], "points_counted": int(counts[2].sum())}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    have_gpu = torch.cuda.is_available()
    dist = None
    if world > 1:
        import torch.distributed as dist
    if args.dry_run:
        dev = torch.device("cuda", local_rank) if have_gpu else torch.device("cpu")
        if world > 1:
            dist.init_process_group("nccl" if have_gpu else "gloo", **(dict(device_id=dev) if have_gpu else {}))
            dist.barrier()
        t = torch.tensor([1.0 + rank], dtype=torch.float64, device=dev)
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        extra = {}
        if args.shard:  # the --shard plumbing without the model: partition, per-rank loads, counter all-reduce
            from cdsegnet_amd import dist as cdist
            sizes = shard_sizes(args)
            mine = cdist.shard_scenes(sizes, rank, world)
            seen = torch.zeros(args.shard, dtype=torch.int64, device=dev)
            seen[mine] = 1
            load = torch.zeros(world, dtype=torch.int64, device=dev)
            load[rank] = sum(sizes[i] for i in mine)
            counts = torch.zeros(3, 16, dtype=torch.int64, device=dev)
            counts[2, rank % 16] = load[rank]
            if world > 1:
                dist.all_reduce(seen, op=dist.ReduceOp.SUM)
                dist.all_reduce(load, op=dist.ReduceOp.SUM)
            cdist.reduce_counts(counts)
            extra = {"shard_scenes": args.shard, "every_scene_on_exactly_one_rank": bool((seen == 1).all()),
                     "points_per_rank": load.tolist(), "points_total": int(sum(sizes)), "points_counted": int(counts[2].sum())}
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int(ones.item()), "max_over_ranks": float(t.item()),
                              "backend": "nccl" if have_gpu else "gloo", **extra}))
        if world > 1:
            dist.destroy_process_group()
        return
    assert have_gpu, "bench.py needs an MI355X"
    # host side of a rank: NUMA-local CPUs of this rank's GPU, a cap on torch's intra-op threads, blocking host
    # reads - applied when there is more than one rank (a single rank keeps the process as the driver started it), reported
    # in the line either way
    from cdsegnet_amd import dist as cdist0
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    host_plan, host_applied = cdist0.setup_rank_host(local_rank, local_world, apply=world > 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from cdsegnet_amd import _lib, configs, ops
    from cdsegnet_amd import dist as cdist
    from cdsegnet_amd.param_init import fill_state_dict
    from cdsegnet_amd.registry import build_model
    import cdsegnet_amd.models  # noqa: F401

    cfg = configs.cdsegnet_config(args.dataset)
    model = build_model(cfg)
    sd = None
    if rank == 0:
        sd = fill_state_dict(model.state_dict(), seed=0)  # random-init weights of the named architecture
        model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval()
    low = args.precision != "fp32"
    variant = "f16" if args.precision.startswith("fp16") else "bf16"  # which build of the library this run calls
    T = ops.LP_DTYPES[variant] if low else torch.float32
    _lib.activate(variant)  # this thread's direct op calls (kernel timer) go to the engine's build
    if world > 1:  # weights: one RCCL broadcast from rank 0 in the compute dtype (replaces the DDP-ctor broadcast)
        cdist.broadcast_model(model, src=0, weight_dtype=T if T != torch.float32 else None)
    model.precision = args.precision
    model.noise_source = "device"  # noise-branch input drawn by the Philox kernel (no host RNG + PCIe in the step)

    if args.protocol == "paper":
        return paper_protocol(args, model, cfg, dev, rank, world, dist)
    if args.shard:
        return shard_mode(args, model, cfg, dev, rank, world, dist)
    scenes_per_step = args.scenes_per_forward * args.lanes
    scenes = make_scenes(args, rank, scenes_per_step)
    dicts = []
    for sc in scenes:
        d = {k: torch.as_tensor(sc[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset")}
        d["offset_host"] = [int(v) for v in sc["offset"]]
        dicts.append(d)
    sizes = [len(sc["coord"]) for sc in scenes]
    pts_per_step = int(sum(sizes))
    torch.manual_seed(54421566 + rank)

    def run(k):
        """k steps; one step = the batch of `scenes_per_step` distinct scenes through inference_many: one collated
        forward of --scenes-per-forward scenes per lane (no host sync between steps: the lanes keep streaming)."""
        out = None
        for _ in range(k):
            out = model.inference_many([dict(d) for d in dicts], lanes=args.lanes, batch=args.scenes_per_forward)
        return out

    timer = not args.no_kernel_timer
    if args.serial:
        model.engine().fork_stage = None
    out = None
    if args.warmup:
        out = run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    last = out[-1]["seg_logits"]
    assert torch.isfinite(last).all()
    hbm_peak_gb = torch.cuda.max_memory_allocated(dev) / 2 ** 30  # (allocator high-water mark through warm-up + timed region)
    # rounds 1 - 5 timed 8 scenes per forward x 3 lanes (24 scenes per step); round 6 collates 24 per forward (the deep stages'
    # launches are three times as full, profiles/r06_lanes_sweep.txt).  The old setting on this step's first 24 scenes, right
    # after the timed region, so that the line stays comparable with the earlier rounds' `value`
    r5cfg = None
    if (rank == 0 and world == 1 and (args.scenes_per_forward, args.lanes) != (8, 3) and len(dicts) >= 24
            and not args.no_kernel_timer):  # (profiling / A-B runs pass --no-kernel-timer: the timed configuration only)
        sub24 = dicts[:24]
        k5 = max(3, args.steps // 2)
        for _ in range(2):
            model.inference_many([dict(d) for d in sub24], lanes=3, batch=8)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for _ in range(k5):
            model.inference_many([dict(d) for d in sub24], lanes=3, batch=8)
        torch.cuda.synchronize()
        e5 = time.perf_counter() - t5
        r5cfg = {"points_per_s": float(sum(sizes[:24])) * k5 / e5, "ms_per_step": 1e3 * e5 / k5, "steps": k5,
                 "scenes_per_forward": 8, "forwards_in_flight_per_gpu": 3, "scenes_per_step": 24,
                 "what": "the configuration rounds 1 - 5 quoted `value` on (8 scenes collated per forward, 3 forwards in flight), "
                         "the first 24 scenes of this step, timed right after the headline region"}

    # ---- kernel-level pass (rank 0): the SAME forward the timed region issues (first lane's 8 collated scenes), one
    # forward at a time, HIP events around every attention / sparse-conv launch on the launch stream.  Inside the timed
    # region three forwards share the CUs, so a launch's wall time there is not a property of the kernel (and the
    # rocprofv3 --kernel-trace profile of the default run overlaps the lanes as well: profiles/r03_bench_kernel_stats.txt is for
    # the SHARES; the trace of a one-lane run, profiles/r03_bench_lanes1_kernel_stats.txt, is the one whose average
    # durations this pass has to agree with).
    iso = None
    eng = model.engine()
    if timer and rank == 0:
        from cdsegnet_amd.models import collate_device
        torch.cuda.synchronize()
        torch.cuda.empty_cache()  # the lanes' allocator pools would otherwise starve the default stream's
        fwd = collate_device([dict(d) for d in dicts[:args.scenes_per_forward]])
        fork, eng.fork_stage = eng.fork_stage, None
        reps = 5
        try:
            for _ in range(2):
                model.inference(dict(fwd), eval=False)
            torch.cuda.synchronize()
            a0, c0, ab0, cd0 = eng.attn_work, eng.conv_bytes, eng.attn_bytes, eng.conv_deep_bytes
            per, walls = [], []
            for attempt in range(3):  # the five forwards must agree (max / min of the attention totals <= 1.15): a clock
                per, walls = [], []   # dip or an allocator stall inside one of them is a reason to measure again
                for _ in range(reps):
                    ops.attention_prof_enable(True)
                    t1 = time.perf_counter()
                    model.inference(dict(fwd), eval=False)
                    torch.cuda.synchronize()
                    walls.append(1e3 * (time.perf_counter() - t1))
                    per.append(ops.prof_summary(ops.PROF_ATTENTION) + ops.prof_summary(ops.PROF_CONV) +
                               ops.prof_summary(ops.PROF_CONV_DEEP))
                ops.attention_prof_enable(False)
                am = [r[0] for r in per]
                if max(am) <= 1.15 * min(am):
                    break
            nrun = (attempt + 1) * reps
            ams, al = sorted(per, key=lambda r: r[0])[reps // 2][:2]
            cms, cl = sorted(per, key=lambda r: r[2])[reps // 2][2:4]
            dms, dl = sorted(per, key=lambda r: r[4])[reps // 2][4:6]
            am = sorted(r[0] for r in per)
            iso = dict(reps=1, attn_ms=ams, attn_launches=al, attn_work=(eng.attn_work - a0) / nrun, conv_ms=cms,
                       conv_launches=cl, conv_bytes=(eng.conv_bytes - c0) / nrun, attn_bytes=(eng.attn_bytes - ab0) / nrun,
                       deep_ms=dms, deep_launches=dl, deep_bytes=(eng.conv_deep_bytes - cd0) / nrun,
                       points=int(sum(sizes[:args.scenes_per_forward])),
                       attn_ms_all=[round(r[0], 3) for r in per], attn_ms_min=am[0], attn_ms_max=am[-1],
                       attn_spread=am[-1] / am[0], attempts=attempt + 1, forward_wall_ms=float(np.median(walls)),
                       work=eng.forward_work(eng.last_plan))
        finally:
            eng.fork_stage = fork
        # host side of a forward: wall and CPU time of the issuing thread from the call to the return of inference()
        # (GPU idle at the start, no synchronisation at the end; the forward's own two host reads are inside) - what one
        # rank's Python thread needs per forward, next to the GPU time of that forward (the scaling risk of
        # 8 ranks on one host is this thread, not a collective)
        hw, hc = [], []
        for _ in range(5):
            torch.cuda.synchronize()
            t1, c1 = time.perf_counter(), time.thread_time()
            model.inference(dict(fwd), eval=False)
            hw.append(1e3 * (time.perf_counter() - t1))
            hc.append(1e3 * (time.thread_time() - c1))
        torch.cuda.synchronize()
        iso["host_issue_ms"], iso["host_cpu_ms"] = float(np.median(hw)), float(np.median(hc))
        # bs = 1 latency (the reference tester's batch size, test.py:99), side-stream fork on
        one = dict(dicts[0])
        for _ in range(3):
            model.inference(dict(one), eval=False)
        torch.cuda.synchronize()
        lat = []
        for _ in range(9):  # one scene at a time, synchronised: the median is the bs = 1 latency
            t1 = time.perf_counter()
            model.inference(dict(one), eval=False)
            torch.cuda.synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
        iso["latency_ms"] = float(np.median(lat))
        iso["latency_points"] = sizes[0]
        # the reference's timing protocol on this step's distinct scenes (tools/test_time.py: one scene at a time, wall
        # clock): 312 inferences, the ScanNet val split's scene count.  `--protocol paper` runs 312 DISTINCT scenes.
        # The dicts handed over are the REFERENCE's (coord, grid_coord, feat, offset - no offset_host hint): every host
        # read of the call is inside the clock, as in `--protocol paper`.
        ref_dicts = [{k: v for k, v in d.items() if k != "offset_host"} for d in dicts]
        torch.cuda.empty_cache()  # the timed region's lane pools would otherwise crowd the default stream's allocator
        for d in ref_dicts[:3]:
            model.inference(dict(d), eval=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(312):  # (the step's distinct scenes, cycled)
            model.inference(dict(ref_dicts[i % len(ref_dicts)]), eval=False)
        torch.cuda.synchronize()
        iso["paper_s"] = time.perf_counter() - t1
        # IEEE-half trunk: one diagnostic forward that counts the activations clamped at +-65504 (engine.count_saturation)
        if args.precision.startswith("fp16"):
            model.count_saturation = True
            model.inference(dict(ref_dicts[0]), eval=False)
            iso["half_saturation"] = dict(clamped=model.engine().saturation_count, checked=model.engine().saturation_checked)
            model.count_saturation = False
        # BASELINE.json's configs[1] names bf16: the same timed region in the bfloat16 build, next to the headline
        if args.precision == "fp16+head" and world == 1:
            model.precision = "bf16+head"
            _lib.activate("bf16")
            run(2)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run(max(3, args.steps // 2))
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            iso["bf16_head"] = dict(points_per_s=pts_per_step * max(3, args.steps // 2) / el2,
                                    ms_per_step=1e3 * el2 / max(3, args.steps // 2), steps=max(3, args.steps // 2))
            if not args.no_agreement:  # ... and what the 8-bit mantissa costs: the same scene and draws as agreement_vs_fp32 below
                iso["bf16_head"]["agreement_vs_fp32"] = agreement_leg(model, dicts[0], sizes[0], cfg, "bf16+head")
            model.precision = args.precision
            _lib.activate(variant)

    # ---- 16-bit accuracy on a bench scene: same draws through the exact-fp32 HIP path (rank 0)
    agreement = None
    if rank == 0 and low and not args.no_agreement:
        agreement = agreement_leg(model, dicts[0], sizes[0], cfg, args.precision)
        model.precision = args.precision
        model.noise_source = "device"
    parity_mode = None
    if rank == 0 and low and not args.no_agreement:
        # throughput of the exact-fp32 mode (inside north_star's 1e-3 logit bound): 8 collated scenes per forward, two
        # forwards in flight (round 6; 4 x 1 before: 9.3 -> 10.9 M points/s, gpurun_out/r06ae_fp32_sweep.txt), the same scenes
        model.precision = "fp32"
        try:
            nf = min(16, len(dicts))
            sub = [dict(d) for d in dicts[:nf]]
            for _ in range(2):
                model.inference_many([dict(d) for d in sub], lanes=2, batch=8)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                model.inference_many([dict(d) for d in sub], lanes=2, batch=8)
            torch.cuda.synchronize()
            el32 = (time.perf_counter() - t1) / 3
            parity_mode = dict(precision="fp32", points_per_s=float(sum(sizes[:nf]) / el32), ms_per_step=1e3 * el32,
                               scenes_per_forward=8, forwards_in_flight=2,
                               note="exact-fp32 MFMA path (v_mfma_f32_16x16x4_f32): within 6e-6 of the reference's CPU logits on the "
                                    "golden fixtures (tests/test_gpu_e2e.py); the 16-bit trunk of the headline line is the IEEE-half build")
            # ... and of "fp32x3" (round 6): the same fp32 engine with every matrix product as three IEEE-half MFMAs on split
            # operands (csrc/gemm.hip, attention.hip) - the fast path inside the 1e-3 bound; 8 collated scenes, 2 lanes
            model.precision = "fp32x3"
            sub8 = [dict(d) for d in dicts[:16]]
            for _ in range(2):
                model.inference_many([dict(d) for d in sub8], lanes=2, batch=8)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                model.inference_many([dict(d) for d in sub8], lanes=2, batch=8)
            torch.cuda.synchronize()
            elx3 = (time.perf_counter() - t1) / 3
            parity_mode["fp32x3"] = dict(precision="fp32x3", points_per_s=float(sum(sizes[:16]) / elx3), ms_per_step=1e3 * elx3,
                                         scenes_per_forward=8, forwards_in_flight=2,
                                         speedup_over_fp32=float(sum(sizes[:16]) / elx3) / parity_mode["points_per_s"],
                                         note="fp32 tensors, split-half operands (x ~= hi + lo' / 2048), three half MFMAs per "
                                              "product, fp32 accumulation; attention scores on half pairs, P V on bfloat16 pairs")
        finally:
            model.precision = args.precision

    # per-class intersection/union/target counters of the last scene: the per-scene record the reference
    # gathers over gloo (test.py:374) - here one RCCL all-reduce (random-init weights: the value is meaningless)
    counts = cdist.confusion_counts(last.argmax(1), torch.as_tensor(scenes[-1]["segment"]).to(dev), last.shape[1])
    cdist.reduce_counts(counts)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    pts = torch.tensor([pts_per_step], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(pts, op=dist.ReduceOp.SUM)
    per_rank_rates = None
    if world > 1:  # every rank's own rate next to the max-over-ranks time the headline uses
        rates = torch.zeros(world, dtype=torch.float64, device=dev)
        rates[rank] = pts_per_step * args.steps / elapsed
        dist.all_reduce(rates, op=dist.ReduceOp.SUM)
        per_rank_rates = [float(v) for v in rates.tolist()]
    elapsed = float(tmax.item())
    total_pts = int(pts.item())

    if rank == 0:
        shape = {"scannet": "ScanNet", "scannet200": "ScanNet200", "nuscenes": "nuScenes"}[args.dataset]
        res = {
            "metric": "points/sec/node (ScanNet ~120k-pt scenes, 1-step)",
            "value": total_pts * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype_name(args.precision),
            "data": "synthetic",
            "config": {"workload": f"{shape}-shape scenes{' after coord noise 0.05 m + 50 % drop + re-voxelisation' if args.robust else ''}"
                                   f", CDSegNet 1-step inference (PT-v3m1 dual backbone, 101.4M params, random-init), "
                                   f"{scenes_per_step} distinct scenes/step/GPU",
                       "points_per_scene_mean": pts_per_step / scenes_per_step, "points_per_scene_min": min(sizes),
                       "points_per_scene_max": max(sizes), "precision": args.precision,
                       "scenes_per_step_per_gpu": scenes_per_step, "scenes_per_forward": args.scenes_per_forward,
                       "forwards_in_flight_per_gpu": args.lanes, "hbm_peak_allocated_gb": round(hbm_peak_gb, 2), "noise": "device Philox",
                       "trunk_16bit_type": ("IEEE half (fp16)" if variant == "f16" else "bfloat16") if low else "none (fp32)",
                       "trunk_16bit_note": "BASELINE.json configs[1] names bf16; the default trunk is IEEE half - the same MFMA rate and "
                                           "bytes as bfloat16, 11 instead of 8 mantissa bits (arg-max agreement with the exact-fp32 path "
                                           "99.94 % instead of 99.5 %; the reference's own GPU path computes its attention in half, "
                                           "ptv3.py:282) - with fp32 residual stream, fp32 accumulation and fp32 heads; "
                                           "--precision bf16+head runs the bfloat16 build; parity_mode = the exact-fp32 path",
                       "host_hints": ["offset_host"],
                       "host_hints_note": "the scene dicts carry offset_host (the batch offsets as Python ints) next to the "
                                          "reference's keys: collating resident scenes into one forward (models.collate_device) "
                                          "then needs no device->host read per scene; the engine itself has not needed the hint "
                                          "since round 6 (the offsets come to the host with the forward's one read)"},
        }
        if iso and iso["attn_ms"] > 0:
            peak = PEAK_TFLOPS[args.precision]
            r = iso["reps"]
            achieved = iso["attn_work"] / (iso["attn_ms"] * 1e-3) / 1e12
            res["roofline"] = {
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                "kernel": ("attn_bf16_kernel" + (" (IEEE-half build: half Q K^T, bfloat16 P V)" if variant == "f16" else "")) if low else "attn_f32_kernel",
                "launches_per_forward": iso["attn_launches"] / r, "avg_launch_us": 1e3 * iso["attn_ms"] / iso["attn_launches"],
                "algorithmic_gflop_per_forward": iso["attn_work"] / r / 1e9, "kernel_ms_per_forward": iso["attn_ms"] / r,
                "algorithmic_bytes_per_launch": iso["attn_bytes"] / max(1, iso["attn_launches"]),
                "scenes_per_forward": args.scenes_per_forward, "points_per_forward": iso["points"],
                "measured": f"HIP events around every launch on the launch stream; the timed configuration's own forward "
                            f"({args.scenes_per_forward} collated scenes), the median of 5 forwards run one at a time right after the "
                            f"timed region (attention ms of the five: {iso['attn_ms_all']})",
                "kernel_ms_min_median_max": [iso["attn_ms_min"], iso["attn_ms"], iso["attn_ms_max"]],
                "spread_max_over_min": iso["attn_spread"], "measurement_attempts": iso["attempts"],
                "stable": bool(iso["attn_spread"] <= 1.15)}
            ceil = attention_ceiling(achieved) if low else None
            if ceil:
                res["roofline"]["frac_of_ceiling"] = ceil.pop("frac_of_ceiling")
                res["roofline"]["ceiling"] = ceil
            wk = iso["work"]
            res["roofline_forward"] = {
                "what": f"one collated forward of {args.scenes_per_forward} scenes, run alone (the roofline pass above)",
                "algorithmic_gflop": wk["total"] / 1e9, "gflop_by_class": {k: v / 1e9 for k, v in wk.items() if k != "total"},
                "wall_ms": iso["forward_wall_ms"], "achieved_tflops": wk["total"] / (iso["forward_wall_ms"] * 1e-3) / 1e12,
                "frac_of_mfma_peak": wk["total"] / (iso["forward_wall_ms"] * 1e-3) / 1e12 / peak,
                "mflop_per_point": wk["total"] / iso["points"] / 1e6,
                "compulsory_bytes": "inputs 60 B / point + logits + the weights once (203 MB in 16 bits): the forward is not HBM-bound as a whole",
                "flops": "SURVEY 8(d) formulas on the plan's real sizes; sparse convs count occupied neighbours only; the dead c-decoder is not run"}
            if iso["conv_ms"] > 0:  # the wide stages' weight-stationary convs: HBM / gather bound
                gbs = iso["conv_bytes"] / (iso["conv_ms"] * 1e-3) / 1e9
                res["roofline_conv"] = {
                    "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                    "kernel": "conv_ll_kernel<32 | 64> (k = 3 sparse convs of the C <= 64 stages; in practice bound by the texture "
                              "path's per-lane tag look-ups of the gathered rows, DESIGN 4.2, profiles/r06_pmc_conv64.txt)",
                    "launches_per_forward": iso["conv_launches"] / r, "kernel_ms_per_forward": iso["conv_ms"] / r,
                    "avg_launch_us": 1e3 * iso["conv_ms"] / max(1, iso["conv_launches"]),
                    "algorithmic_mb_per_forward": iso["conv_bytes"] / r / 1e6,
                    "bytes": "features in + out, kernel map as stored (27 x int32 per point), weights once"}
            if iso.get("deep_ms", 0) > 0:  # the C >= 128 convs on the gathered GEMM: MFMA / LDS-DMA bound
                tf = wk["conv_deep"] / (iso["deep_ms"] * 1e-3) / 1e12
                res["roofline_conv_deep"] = {
                    "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
                    "kernel": "gemm_dma_kernel<128 | 256, GATHER> (k = 3 sparse convs of the C >= 128 stages: 128 x 128 tiles, "
                              "C >= 256: 256 x 256 tiles of 8 waves; bound by LDS-DMA pieces per FLOP and the matrix pipe, DESIGN 4.2)",
                    "launches_per_forward": iso["deep_launches"] / r, "kernel_ms_per_forward": iso["deep_ms"] / r,
                    "avg_launch_us": 1e3 * iso["deep_ms"] / max(1, iso["deep_launches"]),
                    "algorithmic_gflop_per_forward": wk["conv_deep"] / 1e9,
                    "flops": "2 x occupied neighbours x C^2 per row (zero padding of 16-row groups not counted)"}
            res["single_scene_latency_ms"] = iso["latency_ms"]
            res["single_scene_points"] = iso["latency_points"]
            res["paper_protocol"] = {
                "seconds_for_312_scenes": iso["paper_s"], "scenes": 312, "distinct_scenes": len(dicts),
                "input": "the reference's dict (coord, grid_coord, feat, offset): no offset_host hint",
                "protocol": "one scene at a time (bs = 1), no TTA, wall clock incl. every host sync - the reference's "
                            "tools/test_time.py; its published figure for the 312-scene ScanNet val split is 56 s on an RTX 3090 "
                            "(BASELINE.md; other hardware, real scans, data loading excluded there too)",
                "points_per_scene_mean": pts_per_step / scenes_per_step}
            if world == 1 and not args.no_paper_pass and args.dataset == "scannet" and not args.robust:
                # the reference's protocol as its own process (312 DISTINCT scenes, nothing else run before): what
                # `python bench.py --protocol paper` prints, here so that the driver's line carries it
                try:
                    pp = subprocess.run([sys.executable, os.path.abspath(__file__), "--protocol", "paper", "--precision", args.precision,
                                         "--points", str(args.points)], capture_output=True, text=True, timeout=600)
                    pj = json.loads(pp.stdout.strip().splitlines()[-1])
                    res["paper_protocol"]["own_process"] = {"seconds_for_312_scenes": pj["value"], "ms_per_scene": pj["ms_per_step"],
                                                            "points_per_s": pj["points_per_s"], "distinct_scenes": 312,
                                                            "what": "python bench.py --protocol paper in a process of its own, run "
                                                                    "from this one after the timed region"}
                except Exception as e:  # noqa: BLE001 - the headline line must not depend on this leg
                    res["paper_protocol"]["own_process"] = {"error": repr(e)[:200]}
            if "half_saturation" in iso:
                res["half_saturation"] = dict(iso["half_saturation"], what="16-bit activations that reach memory at +-65504, "
                                              "the clamp value of the half build's conversions, in one forward of a bench scene "
                                              "(0 = the trunk stayed inside IEEE half's range)")
            if "bf16_head" in iso:
                res["bf16_head"] = dict(iso["bf16_head"], what="the same timed region with precision bf16+head (bfloat16 build "
                                        "of the library; BASELINE.json configs[1] names bf16), run right after the headline")
            tpath = next((p for p in (os.path.join(ROOT, "profiles", f"r0{k}_attention_traffic.json") for k in (6, 5, 4, 3))
                          if os.path.exists(p)), "")
            if low and os.path.exists(tpath):
                # HBM bytes per launch (mean over every attention launch of this bench's forwards) from separate rocprofv3
                # --pmc passes, FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE: tools/pmc_bench_traffic.sh, an OFFLINE
                # measurement of this build - the live run cannot collect PMCs
                with open(tpath) as f:
                    tj = json.load(f)
                res["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
                res["roofline"]["traffic_source"] = f"profiles/{os.path.basename(tpath)} (offline rocprofv3 --pmc passes over bench.py's own forwards)"
                # the counters belong to the build they were taken on: stale once the kernel's source is newer than the file
                # (by content hash when the profile carries one - file times do not survive a checkout -, else by file time)
                src = os.path.join(ROOT, "cdsegnet_amd", "csrc", "attention.hip")
                if tj.get("attention_hip_sha256"):
                    import hashlib
                    with open(src, "rb") as f:
                        res["roofline"]["traffic_stale"] = hashlib.sha256(f.read()).hexdigest() != tj["attention_hip_sha256"]
                else:
                    res["roofline"]["traffic_stale"] = bool(os.path.getmtime(src) > os.path.getmtime(tpath))
        if r5cfg is not None:
            res["value_at_round5_config"] = r5cfg
        if iso and "latency_ms" in iso:
            # the headline block a reader should see first: `value` needs this build's own `inference_many` (8 scenes collated
            # per forward, 3 forwards in flight) and the offset_host hint; the REFERENCE's loop (tools/test_*.py unchanged, one
            # `inference()` per fragment, engines/test.py:197-224) gets the bs = 1 figures below
            own = res.get("paper_protocol", {}).get("own_process", {})
            res["headline"] = {
                "value_points_per_s": res["value"],
                "value_needs": f"DefaultSegmentorV2.inference_many(batch={args.scenes_per_forward}, lanes={args.lanes}) + the offset_host key (INTEGRATION.md)",
                "bs1_ms_per_scene": own.get("ms_per_scene", 1e3 * iso["paper_s"] / 312.0),
                "bs1_points_per_s": own.get("points_per_s", pts_per_step / scenes_per_step * 312.0 / iso["paper_s"]),
                "bs1_what": "the reference's calling pattern: one inference(dict) per scene, the reference's dict, every host "
                            "sync inside the clock (paper_protocol" + (".own_process" if "ms_per_scene" in own else "") + ")",
                "bs1_over_value": own.get("points_per_s", pts_per_step / scenes_per_step * 312.0 / iso["paper_s"]) / res["value"]}
        if iso and "host_issue_ms" in iso:
            res["host_issue"] = {
                "host_issue_ms_per_forward": iso["host_issue_ms"], "host_cpu_ms_per_forward": iso["host_cpu_ms"],
                "host_issue_ms_per_step": iso["host_issue_ms"] * args.lanes,
                "host_duty": iso["host_issue_ms"] * args.lanes / (1e3 * elapsed / args.steps),
                "host_threads": os.cpu_count(),
                "what": "one Python thread issues every launch of a rank (no data-path collective): wall / CPU time of that "
                        "thread per collated forward, GPU idle at the start, no synchronisation at the end; host_duty = issue "
                        "time of a step's forwards / the step's GPU-bound wall time.  N ranks need N such threads"}
        res["host_plan"] = {
            "rank0": {"numa_node": host_plan["numa_node"], "cpus": len(host_plan["cpus"]), "first_cpu": host_plan["cpus"][0],
                      "last_cpu": host_plan["cpus"][-1], "threads": host_plan["threads"], "blocking_sync": host_plan["blocking_sync"]},
            "applied": host_applied,
            "what": "per rank: CPUs of its GPU's NUMA node (split among the ranks on that node), torch intra-op thread cap, "
                    "hipDeviceScheduleBlockingSync - cdsegnet_amd.dist.rank_host_plan; applied when n_gpus > 1"}
        if world > 1:
            res["per_rank_points_per_s"] = per_rank_rates
            res["slowest_rank_ms_per_step"] = 1e3 * elapsed / args.steps
        if agreement:
            res["agreement_vs_fp32"] = agreement
        if parity_mode:
            res["parity_mode"] = parity_mode
        m = cdist.metrics(counts)
        res["eval_counters"] = {"mIoU_random_init": m["mIoU"], "points_counted": int(counts[2].sum())}
        if args.cpu_baseline and world == 1:
            res["cpu_baseline"], cpu_case = cpu_baseline(cfg, sd, args.cpu_points, args.dataset, args.cpu_threads)
            if low:
                live = parity_vs_cpu(model, dev, cpu_case, ["fp32", "fp32x3", args.precision])
                res.setdefault("parity_mode", {"precision": "fp32"}).update(live["fp32"])
                if isinstance(res["parity_mode"].get("fp32x3"), dict):
                    res["parity_mode"]["fp32x3"].update(live["fp32x3"])
                res["parity_vs_cpu_oracle"] = {"scene_points": int(cpu_case[2].shape[0]), **{k: v for k, v in live.items()}}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

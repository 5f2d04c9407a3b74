"""What does the HOST cost per forward when N ranks issue at the same time?  (evidence for DESIGN.md 6 without an 8-GPU node)
   python tools/host_contention.py [ranks=8] [points=40000] [scenes_per_forward=8] [forwards=20]
N processes - one per would-be GPU rank - each build the full-width model and issue collated forwards concurrently, all on
the ONE GPU of the box (they time-share it: wall time per forward is meaningless here and not the point).  What an 8-GPU
node shares between its ranks is the host: this measures the CPU time of the issuing Python thread per forward
(time.thread_time: scheduler-independent; host reads block instead of spinning - hipDeviceScheduleBlockingSync - so the
time the thread sleeps waiting for the time-shared GPU is not counted) and its wall time from call to return, alone
(N = 1) and contended (N ranks), plus the node's core count.  There is no collective in the step, so nothing else couples
the ranks."""
import json
import os
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, points, spf, forwards, barrier, q):
    import ctypes
    import numpy as np
    # host reads BLOCK instead of spinning (hipDeviceScheduleBlockingSync, set before the first HIP call of the process):
    # with the default policy the thread burns CPU inside the forward's two device->host reads for as long as the
    # (time-shared) GPU makes it wait, and thread_time would measure the GPU's queue, not the host's work
    if os.environ.get("CDSEG_HOST_PLAN"):
        # the shipped per-rank host plan (dist.setup_rank_host, what bench.py applies for WORLD_SIZE > 1): CPU slice, thread cap
        # and blocking sync in one call - on a one-GPU box every rank falls back to a slice of all allowed CPUs
        from cdsegnet_amd import dist as cdist
        # (plan only, then applied by hand: setup_rank_host(apply=True) would select GPU `rank`, and this box has one)
        plan, _ = cdist.setup_rank_host(rank, world, apply=False)
        plan["blocking_sync"] = True
        applied = cdist.apply_rank_host_plan(plan)
        assert applied["blocking_sync"], applied
        if rank == 0:
            print(f"# rank 0 of {world}: {applied}", flush=True)
    else:
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipSetDeviceFlags(ctypes.c_uint(0x4))
        assert rc == 0, f"hipSetDeviceFlags failed: {rc}"
    from cdsegnet_amd import configs, synth
    from cdsegnet_amd.models import collate_device
    from cdsegnet_amd.param_init import fill_state_dict
    from cdsegnet_amd.registry import build_model
    import cdsegnet_amd.models  # noqa: F401
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cfg = configs.cdsegnet_config("scannet")
    model = build_model(cfg)
    model.load_state_dict(fill_state_dict(model.state_dict(), seed=0))
    model = model.to(dev).eval()
    model.precision = "fp16+head"
    model.noise_source = "device"
    dicts = []
    for i in range(spf):
        sc = synth.room_scene(1000 * rank + i, points)
        d = {k: torch.as_tensor(sc[k]).to(dev) for k in ("coord", "grid_coord", "feat", "offset")}
        d["offset_host"] = [int(v) for v in sc["offset"]]
        dicts.append(d)
    torch.manual_seed(54421566 + rank)
    eng = model.engine()
    eng.fork_stage = None
    for _ in range(3):
        model.inference(collate_device([dict(d) for d in dicts]), eval=False)
    torch.cuda.synchronize()
    barrier.wait()
    cpu, wall = [], []
    t_all = time.perf_counter()
    for _ in range(forwards):
        t0, c0 = time.perf_counter(), time.thread_time()
        model.inference(collate_device([dict(d) for d in dicts]), eval=False)
        wall.append(1e3 * (time.perf_counter() - t0))
        cpu.append(1e3 * (time.thread_time() - c0))
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    q.put(dict(rank=rank, host_cpu_ms_per_forward=float(np.median(cpu)), host_issue_wall_ms_per_forward=float(np.median(wall)),
               host_cpu_ms_p90=float(np.percentile(cpu, 90)), ms_per_forward_incl_gpu_time_sharing=1e3 * total / forwards))


def run(world, points, spf, forwards):
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(world), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, points, spf, forwards, barrier, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join()
    return sorted(res, key=lambda r: r["rank"])


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    points = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
    spf = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    forwards = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    print(f"# host threads {os.cpu_count()}, {spf} collated scenes of {points} voxels per forward, {forwards} forwards per rank, "
          f"all ranks on GPU 0 (time-shared)")
    for w in sorted({1, world}):
        res = run(w, points, spf, forwards)
        cpu = [r["host_cpu_ms_per_forward"] for r in res]
        wl = [r["host_issue_wall_ms_per_forward"] for r in res]
        print(f"ranks={w}: host_cpu_ms_per_forward median over ranks {sorted(cpu)[len(cpu) // 2]:.2f} (min {min(cpu):.2f}, max {max(cpu):.2f}); "
              f"issue wall ms per forward median {sorted(wl)[len(wl) // 2]:.2f} (max {max(wl):.2f})")
        for r in res:
            print("   " + json.dumps(r))

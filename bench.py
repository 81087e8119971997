#!/usr/bin/env python3
"""bench.py -- the headline measurement: Mrays/s of the isosurface ray caster on the Skull iso sweep.

One step = one pass of the hot path over one batch of work: the reference driver's iso loop
(src/main.zig:111-138: iso = 0..255, one renderImage each) at 1024 x 1024 on Skull.vol with the
reference camera (src/main.zig:31-43), i.e. 256 x 1 048 576 primary rays per step.  All primary
rays count, including the ones that miss the volume -- the reference's ms/frame includes them too.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--gather]
                    [--kernel V] [--width PX] [--no-cpu-baseline]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py`
(one process per GPU, RANK/LOCAL_RANK/WORLD_SIZE from the environment).  The path shards with no
data-path collective: every rank holds a replica of the node array.
  weak   (default) each rank renders the whole 256-iso sweep; the job is N sweeps
  strong the 256 iso values of ONE sweep are split into N contiguous blocks (north-star sharding)
--gather adds the one real exchange step: framebuffers are gathered to rank 0 over RCCL inside the
timed region.

Inputs (node array) and outputs (RGB24 frames) are resident in HBM across the timed region; the
PCIe copy of the frames to the host is not part of `value`.

Rank 0 prints ONE JSON line.  `roofline` prices the traversal kernel against the HBM peak with the
ALGORITHMIC bytes 4 * nodes_checked + 3 * pixels (one 4-byte SlimNode per reference loop iteration,
src/raytrace.zig:162,177, plus one RGB24 store per pixel, :54-67), its duration measured with HIP
events on the launch stream.  `cpu_baseline` times the CPU oracle (a port of the reference's
single-threaded path; the Zig reference cannot be built offline) on one host core on a bounded
sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def cpu_baseline(kdt, pos, forw, up, width, budget_s=20.0):
    """Times the oracle's render_image (1 core) on every 4th iso of the sweep until the budget is spent."""
    from oracle import oracle as O
    isos = list(range(0, 256, 4))
    done, rays, checks = [], 0, 0
    t0 = time.perf_counter()
    for iso in isos:
        r = O.render(kdt.nodes, kdt.root_dim, pos, forw, up, width, width, 2.0, iso, taps=False)
        done.append(iso)
        rays += width * width
        checks += r["stats"].nodes_checked
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
            "sample": "Skull %dx%d, isos %s (%d frames, %.1f s, tree build excluded), CPU: %s" % (
                width, width, ",".join(map(str, done)), len(done), dt, model),
            "nodes_checked_per_frame": checks / max(1, len(done))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--gather", action="store_true")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import volumerenderer_amd as vr
    from volumerenderer_amd import capi, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d"
                             % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W = H = args.width
    data = vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz"))
    kdt = vr.SlimKDT.newHeursiticPartition(data, "AlwaysCenter")
    pos, forw, up = vr.main_camera(data.resulution)

    L = capi.lib()
    r = vr.Renderer(local_rank)
    r.upload_kdt(kdt)
    r.set_kernel(args.kernel)
    if args.scaling == "strong":
        iso_begin, iso_count = sharding.iso_shard(rank, world, 0, 256)
        counts = [sharding.iso_shard(k, world, 0, 256)[1] for k in range(world)]
    else:
        iso_begin, iso_count = 0, 256
        counts = [256] * world
    view = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    d_rgb = torch.empty((iso_count, H, W, 3), dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((iso_count, 7), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        capi.check(L.vr_render_device(r.handle, C.byref(view), iso_begin, iso_count, 0, H, d_rgb.data_ptr(),
                                      d_stats.data_ptr(), None, C.c_void_p(stream)), "vr_render_device")
        if args.gather and world > 1:
            sharding.gather_frames(d_rgb, counts, dst=0)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    capi.check(L.vr_set_timing(r.handle, 1), "vr_set_timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    capi.check(L.vr_set_timing(r.handle, 0), "vr_set_timing")
    tot_ms, launches = C.c_double(), C.c_uint32()
    capi.check(L.vr_timing_read(r.handle, C.byref(tot_ms), C.byref(launches)), "vr_timing_read")

    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    stats = d_stats.cpu().numpy()
    nodes_checked = int(stats[:, 0].sum())
    errors = int(stats[:, 6].sum())
    rays_per_step_rank = iso_count * W * H
    rays_total = torch.tensor([rays_per_step_rank], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(rays_total, op=dist.ReduceOp.SUM)
    rays_per_step = float(rays_total.item())

    if rank == 0:
        kernel_ms = tot_ms.value / max(1, launches.value)
        alg_bytes = 4.0 * nodes_checked + 3.0 * rays_per_step_rank
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        out = {
            "metric": "Mrays/s on the Skull.vol iso sweep (iso 0..255) at %dx%d" % (W, H),
            "value": rays_per_step * args.steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "Skull.vol shipped by the reference (68x256x256 u8), reference camera; no synthetic stand-in",
            "config": {"workload": "skull_%dx%d_iso0-255_sweep" % (W, H), "volume": "Skull 68x256x256",
                       "tree": "SlimKDT AlwaysCenter, 2^24 slots", "rays_per_step": int(rays_per_step),
                       "sharding": "none" if world == 1 else ("iso-blocks" if args.scaling == "strong" else
                                                              "one sweep per rank"),
                       "gather": bool(args.gather and world > 1),
                       "kernel": capi.lib().vr_kernel_name(args.kernel).decode()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "vr_k_trace" if args.kernel in (0, 4) else "vr_k_render", "kernel_ms": kernel_ms, "launches_timed": int(launches.value),
                         "algorithmic_bytes_per_launch": alg_bytes, "nodes_checked_per_launch": nodes_checked},
            "errors": errors,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(kdt, pos, forw, up, W)
            out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

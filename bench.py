#!/usr/bin/env python3
"""bench.py -- the headline measurement: Mrays/s of the isosurface ray caster on the Skull iso sweep.

One step = one pass of the hot path over one batch of work: the reference driver's iso loop
(src/main.zig:111-138: iso = 0..255, one renderImage each) at 1024 x 1024 on Skull.vol with the
reference camera (src/main.zig:31-43), i.e. 256 x 1 048 576 primary rays per step.  All primary
rays count, including the ones that miss the volume -- the reference's ms/frame includes them too.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong] [--gather] [--in-flight F]
                    [--shard iso|rows] [--kernel V] [--width PX] [--volume skull|c60|standin256|gyroid1024]
                    [--iso-begin B] [--iso-count N] [--no-cpu-baseline]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py`
(one process per GPU, RANK/LOCAL_RANK/WORLD_SIZE from the environment).  The path shards with no
data-path collective: every rank holds a replica of the node array.
  weak   (default) each rank renders the whole 256-iso sweep; the job is N sweeps
  strong the 256 iso values of ONE sweep are split into N contiguous blocks (north-star sharding), or,
         with --shard rows, the rows of every frame
--gather adds the one real exchange step: framebuffers are gathered to rank 0 over RCCL inside the
timed region.

Inputs (node array) and outputs (RGB24 frames) are resident in HBM across the timed region; the
PCIe copy of the frames to the host is not part of `value`.

Rank 0 prints ONE JSON line.  `roofline` prices the traversal kernel against the HBM peak with the
ALGORITHMIC bytes 4 * nodes_checked + 3 * pixels (one 4-byte SlimNode per reference loop iteration,
src/raytrace.zig:162,177, plus one RGB24 store per pixel, :54-67), its duration measured with HIP
events on the launch stream.  `cpu_baseline` times the CPU oracle (a port of the reference's
single-threaded path; the Zig reference cannot be built offline) on one host core on a bounded
sample of the same workload; `cpu_all_cores` is the same oracle with the image rows spread over the
host cores this job may use -- a whole-socket figure the reference itself does not have.
`roofline.measured_copy_GBps` is a device-to-device copy timed in the same process (the HBM ceiling
actually reachable, next to the nominal peak); `roofline.counters` repeats the VALU figures of the
committed rocprofv3 counter pass.
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


def cpu_baseline(kdt, pos, forw, up, width, iso_begin=0, iso_count=256, budget_s=20.0):
    """Times the oracle's render_image (1 core) on every 4th iso of the sweep until the budget is spent."""
    from oracle import oracle as O
    isos = list(range(iso_begin, iso_begin + iso_count, 4 if iso_count >= 8 else 1))
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
            "sample": "%dx%d, isos %s (%d frames, %.1f s, tree build excluded), CPU: %s" % (
                width, width, ",".join(map(str, done)), len(done), dt, model),
            "nodes_checked_per_frame": checks / max(1, len(done))}


def cpu_all_cores(kdt, pos, forw, up, width, threads, iso_begin=0, iso_count=256, budget_s=8.0):
    """The oracle with the rows of each frame spread over `threads` host threads (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from oracle import oracle as O
    bands = [(width * i // (4 * threads), width * (i + 1) // (4 * threads)) for i in range(4 * threads)]
    done, rays = [], 0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        for iso in range(iso_begin + (1 if iso_count > 1 else 0), iso_begin + iso_count, 2):
            img = np.zeros((width, width, 3), np.uint8)
            list(pool.map(lambda b: O.render(kdt.nodes, kdt.root_dim, pos, forw, up, width, width, 2.0, iso, rows=b,
                                             taps=False, out=img), bands))
            done.append(iso)
            rays += width * width
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port, rows over threads (not the reference)",
            "sample": "%d frames (isos %d..%d step 2) in %.1f s" % (len(done), done[0], done[-1], dt)}


def copy_ceiling_gbps(torch):
    """Device-to-device copy of 1 GiB, read + write bytes over the best of 5 timings."""
    a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    best = 1e9
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del a, b
    return 2.0 * (1 << 30) / (best * 1e-3) / 1e9


def profile_counters():
    """VALU figures of the committed counter pass (profiles/r01_pmc_summary.json), for the default workload."""
    try:
        s = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
        k = [v for n, v in s.items() if "vr_k_sweep" in n][0]
        simd_quanta = k["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0  # cycles per XCD / 4-cycle issue slots x 1024 SIMDs
        return {"valu_busy_frac": k["SQ_ACTIVE_INST_VALU"] / simd_quanta,
                "valu_lane_util_frac": k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_ACTIVE_INST_VALU"]),
                "waves_per_simd": k["SQ_WAVES"] / 1024.0, "source": "profiles/r01_pmc_summary.json"}
    except (OSError, ValueError, KeyError, IndexError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--shard", choices=["iso", "rows"], default="iso",
                    help="with --scaling strong: split the iso values (default) or the image rows over the ranks")
    ap.add_argument("--gather", action="store_true")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--volume", choices=["skull", "c60", "standin256", "gyroid1024"], default="skull")
    ap.add_argument("--iso-begin", type=int, default=0)
    ap.add_argument("--iso-count", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="independent steps in flight per rank (own context, stream and frame buffer each): with 2, the "
                         "tail of one sweep, where its last long rays leave CUs idle, overlaps the start of the next; "
                         "the default 1 runs the steps strictly one after the other")
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
    # VR_BENCH_REHEARSAL=1: rehearse the N > 1 control flow on a box with ONE GPU -- every rank renders on device 0,
    # the process group is gloo and the reductions run on host tensors.  Its numbers mean nothing (the ranks share a GPU).
    rehearsal = os.environ.get("VR_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    red_dev = "cpu" if rehearsal else "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W = H = args.width
    if args.volume == "skull":
        data = vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "Skull.vol.gz"))
        what = "Skull.vol shipped by the reference (68x256x256 u8), reference camera; no synthetic stand-in"
    elif args.volume == "c60":
        data = vr.Data.loadUCD(os.path.join(ROOT, "tests", "data", "C60.vol"))
        what = "C60.vol shipped by the reference (64^3 u8), reference camera"
    elif args.volume == "gyroid1024":
        # beyond the reference's u8 coordinates: wide node words, 2^31-slot tree (8 GiB) built on the GPU
        data = vr.Data(vr.datasets.gyroid_noise(1024))
        what = "synthetic 1024^3 u8 gyroid shell + uniform noise (seed 42, volumerenderer_amd/datasets.py), reference camera"
    else:
        data = vr.Data(vr.datasets.bonsai_standin())
        what = "synthetic 256^3 u8 stand-in for the 256^3 CT volume the reference does not ship (seeded, volumerenderer_amd/datasets.py)"
    wide = max(data.resulution) > 256
    I0, IN = args.iso_begin, args.iso_count
    if I0 < 0 or IN < 1 or I0 + IN > 256:
        raise SystemExit("--iso-begin/--iso-count must stay inside 0..255")
    pos, forw, up = vr.main_camera(data.resulution)

    L = capi.lib()
    r = vr.Renderer(local_rank)
    r.upload_volume(data)
    tb = time.perf_counter()
    build = r.build_kdt("AlwaysCenter")  # centre-split tree built on the GPU from the uploaded volume
    build_ms = (time.perf_counter() - tb) * 1e3
    r.set_kernel(args.kernel)
    row_begin, row_end = 0, H
    if args.scaling == "strong" and args.shard == "rows":
        if args.gather:
            raise SystemExit("--gather goes with iso sharding")
        iso_begin, iso_count = I0, IN
        counts = [IN] * world
        row_begin, row_end = sharding.row_shard(rank, world, H)
    elif args.scaling == "strong":
        iso_begin, iso_count = sharding.iso_shard(rank, world, I0, IN)
        counts = [sharding.iso_shard(k, world, I0, IN)[1] for k in range(world)]
    else:
        iso_begin, iso_count = I0, IN
        counts = [IN] * world
    view = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    d_rgb = torch.empty((iso_count, row_end - row_begin, W, 3), dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((iso_count, 7), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # --in-flight > 1: further slots, each with its own context (scratch), stream and output buffers
    if args.in_flight < 1 or (args.in_flight > 1 and args.gather):
        raise SystemExit("--in-flight must be >= 1 and does not go with --gather")
    slots = [(r, d_rgb, d_stats, None)]
    for _ in range(1, args.in_flight):
        r2 = vr.Renderer(local_rank)
        r2.upload_volume(data)
        r2.build_kdt("AlwaysCenter")
        r2.set_kernel(args.kernel)
        s2 = torch.cuda.Stream()
        slots.append((r2, torch.empty_like(d_rgb), torch.zeros_like(d_stats), s2))
    step_no = [0]

    def step():
        rr, rgb_k, stats_k, s_k = slots[step_no[0] % len(slots)]
        step_no[0] += 1
        capi.check(L.vr_render_device(rr.handle, C.byref(view), iso_begin, iso_count, row_begin, row_end, rgb_k.data_ptr(),
                                      stats_k.data_ptr(), None, C.c_void_p(s_k.cuda_stream if s_k else stream)),
                   "vr_render_device")
        if args.gather and world > 1:
            sharding.gather_frames(d_rgb.cpu() if rehearsal else d_rgb, counts, dst=0)

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
    if args.in_flight > 1:
        # kernels of overlapping steps wait for each other's CUs, so their event intervals say nothing about the
        # kernel: time it in a short pass of its own, one step after the other (outside the timed region)
        capi.check(L.vr_set_timing(r.handle, 1), "vr_set_timing")
        for _ in range(3):
            step_no[0] = 0
            step()
            torch.cuda.synchronize()
        capi.check(L.vr_set_timing(r.handle, 0), "vr_set_timing")
        capi.check(L.vr_timing_read(r.handle, C.byref(tot_ms), C.byref(launches)), "vr_timing_read")

    t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    stats = d_stats.cpu().numpy()
    nodes_checked = int(stats[:, 0].sum())
    errors = int(stats[:, 6].sum())
    rays_per_step_rank = iso_count * W * (row_end - row_begin)
    rays_total = torch.tensor([rays_per_step_rank], dtype=torch.float64, device=red_dev)
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
            "metric": "Mrays/s on the %s iso sweep (iso %d..%d) at %dx%d" % (
                {"skull": "Skull.vol", "c60": "C60.vol", "standin256": "256^3 stand-in",
                 "gyroid1024": "1024^3 gyroid"}[args.volume], I0, I0 + IN - 1, W, H),
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
            "data": what,
            "config": {"workload": "%s_%dx%d_iso%d-%d_sweep" % (args.volume, W, H, I0, I0 + IN - 1),
                       "volume": "%s %dx%dx%d" % ((args.volume,) + tuple(data.resulution)),
                       "tree": "SlimKDT AlwaysCenter built on the GPU, %d nodes, depth %d" % (
                           build.number_of_nodes, build.layer_count),
                       "rays_per_step": int(rays_per_step),
                       "sharding": "none" if world == 1 else (
                           ("row-blocks" if args.shard == "rows" else "iso-blocks") if args.scaling == "strong" else
                           "one sweep per rank"),
                       "gather": bool(args.gather and world > 1),
                       "kernel": "wide" if wide else capi.lib().vr_kernel_name(args.kernel).decode()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": ("vr_k_render_wide" if args.kernel in (1, 2, 3) else "vr_k_trace_wide") if wide else (
                             {0: "vr_k_sweep" if IN >= 8 else "vr_k_trace", 4: "vr_k_trace", 5: "vr_k_sweep"}.get(
                                 args.kernel, "vr_k_render")), "kernel_ms": kernel_ms, "launches_timed": int(launches.value),
                         "algorithmic_bytes_per_launch": alg_bytes, "nodes_checked_per_launch": nodes_checked,
                         "measured_copy_GBps": copy_ceiling_gbps(torch),
                         "note": ("algorithmic bytes = the reference's node reads (4 B x nodes_checked) + 3 B per pixel; the "
                                  "kernels read far fewer nodes (resume traversal; the sweep kernel also shares traversals "
                                  "between iso values), so frac prices the reference's work and can exceed 1 -- HBM itself "
                                  "moves `traffic` bytes per launch"),
                         "counters": profile_counters() if (args.volume == "skull" and W == 1024 and IN == 256) else None},
            "errors": errors,
            "steps_in_flight": args.in_flight,
        }
        if rehearsal:
            out["rehearsal"] = "all ranks on one GPU over gloo: control-flow check only, not a measurement"
        if args.volume != "skull" or W != 1024 or IN != 256:
            out["roofline"]["traffic"] = None  # the committed counter pass is for the default workload only
        if wide:
            out["build_ms_gpu"] = build_ms  # the host builder and the CPU oracle cannot run at this size
        elif not args.no_cpu_baseline and world == 1:
            kdt = vr.SlimKDT.newHeursiticPartition(data, "AlwaysCenter")  # host copy for the CPU legs
            out["cpu_baseline"] = cpu_baseline(kdt, pos, forw, up, W, I0, IN)
            out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
            threads = max(1, min(16, len(os.sched_getaffinity(0))))
            out["cpu_all_cores"] = cpu_all_cores(kdt, pos, forw, up, W, threads, I0, IN)
        print(json.dumps(out), flush=True)
    for extra in slots[1:]:
        extra[0].close()
    r.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

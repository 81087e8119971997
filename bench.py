#!/usr/bin/env python3
"""bench.py -- the headline measurement: Mrays/s of the isosurface ray caster on the Skull iso sweep.

One step = one pass of the hot path over one batch of work: the reference driver's iso loop
(src/main.zig:111-138: iso = 0..255, one renderImage each) at 1024 x 1024 on Skull.vol with the
reference camera (src/main.zig:31-43), i.e. 256 x 1 048 576 primary rays per step.  All primary
rays count, including the ones that miss the volume -- the reference's ms/frame includes them too.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling auto|weak|strong] [--shard bands|rows|iso]
                    [--gather] [--no-extras] [--in-flight F] [--kernel V] [--width PX]
                    [--volume skull|c60|standin256|gyroid1024] [--iso-begin B] [--iso-count N] [--no-cpu-baseline] [--no-first-call]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py`
(one process per GPU, RANK/LOCAL_RANK/WORLD_SIZE from the environment).  The path shards with no
data-path collective: every rank holds a replica of the node array.
  strong (what `auto` picks) ONE sweep is split over the ranks: blocks of iso values (--shard iso, the default: the
         0..255 sweep of the reference sharded by iso value), the rows of every frame in bands of 32 dealt out round-robin
         (bands), or contiguous blocks of rows (rows)
  weak   each rank renders a whole sweep of its own; the job is N sweeps
For N > 1 the line carries the main figure (`value`, the scaling named in `scaling`) and, unless --no-extras,
the other ones measured in the same run: `weak_scaling` (or `strong_scaling`), `strong_bands`, and `strong_with_gather`
(the split sweep with the framebuffers gathered to rank 0 over RCCL inside the timed region -- the one real exchange step).

Inputs (node array) and outputs (RGB24 frames) are resident in HBM across the timed region; `value_host_delivered`
(N = 1) is the same sweep delivered into page-locked host memory by vr_render_sweep, which is what the
reference's timed call (src/main.zig:119-135) hands its caller -- never part of `value`.

Rank 0 prints ONE JSON line.  `roofline` prices the traversal kernel against the HBM peak with the
ALGORITHMIC bytes 4 * nodes_checked + 3 * pixels (one 4-byte SlimNode per reference loop iteration,
src/raytrace.zig:162,177, plus one RGB24 store per pixel, :54-67), its duration measured with HIP
events on the launch stream.  Next to that contract figure it reports what the kernel really does: the sweep
kernel's own count of traversals and the share of its run time after its queue ran empty (live, from the kernel),
and HBM traffic / VALU lane utilisation from the committed rocprofv3 counter pass of the same workload
(profiles/r03_pmc_<workload>.json), flagged `counters_stale` when the kernel sources have changed since.  Because the
union kernel reads ~45 times fewer nodes than the reference checks, `frac` is a work ratio; the line therefore also carries
the floors a step can be held against: `hbm_floor_ms` (the frames written once plus the kernel's own HBM reads at the HBM
peak), `latency_floor_ms` (the longest single (ray, iso value) walk of the sweep, timed alone with the per-iso kernel: no
cut of the iso range gets a launch below it) and `bound`, which names the larger of the two.
`cpu_baseline` times the CPU oracle (a port of the reference's single-threaded path; the Zig reference cannot be
built offline) on one host core on a bounded sample of the same workload; `cpu_all_cores` is the same oracle
with the image rows spread over the host cores this job may use -- a whole-socket figure the reference itself
does not have.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
KERNEL_SOURCES = ["vr_kernels.hip", "vr_trace_core.h", "vr_trace_wide.h", "vr_trace_union.h", "vr_trace_union_wide.h"]


def kernel_sources_sha():
    """Fingerprint of the kernel sources a counter summary belongs to."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "volumerenderer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


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


def profile_counters(workload, kernel, kernel_ms):
    """HBM traffic and VALU figures of the committed counter pass of this workload, with the fingerprint of the
    kernel sources they were taken on; stale = the sources of the tree this runs in differ."""
    s = k = path = None
    for rnd in ("r03", "r02"):  # the newest committed pass of this workload
        path = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (rnd, workload))
        try:
            s = json.load(open(path))
            k = s["kernels"][kernel]
            break
        except (OSError, ValueError, KeyError):
            s = k = None
    if k is None:
        return None, None, None
    out = {"source": os.path.relpath(path, ROOT), "kernel_sources_sha": s.get("kernel_sources_sha"),
           "commit": s.get("commit")}
    stale = s.get("kernel_sources_sha") != kernel_sources_sha()
    traffic = None
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies the 128-byte fabric read requests at 64 bytes
        # each, so it is doubled; per launch
        traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        out["hbm_read_bytes"] = 2.0 * k["FETCH_SIZE"] * 1024.0
        out["hbm_write_bytes"] = k["WRITE_SIZE"] * 1024.0
    if "SQ_ACTIVE_INST_VALU" in k and "GRBM_GUI_ACTIVE" in k:
        simd_quanta = k["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0  # cycles per XCD / 4-cycle issue slots x 1024 SIMDs
        out["valu_busy_frac"] = k["SQ_ACTIVE_INST_VALU"] / simd_quanta
    if "SQ_THREAD_CYCLES_VALU" in k and "SQ_ACTIVE_INST_VALU" in k:
        out["valu_lane_util_frac"] = k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVES" in k:
        out["waves_per_simd"] = k["SQ_WAVES"] / 1024.0
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in k and "TCP_TCC_READ_REQ_sum" in k and k["TCP_TOTAL_CACHE_ACCESSES_sum"]:
        out["l1_hit_frac"] = 1.0 - k["TCP_TCC_READ_REQ_sum"] / k["TCP_TOTAL_CACHE_ACCESSES_sum"]
    return out, traffic, stale


def longest_walk(L, C, torch, capi, r, view, W, H, I0, IN, stream):
    """The sweep's longest single (ray, iso value) walk, timed alone: every iso value is rendered once by the kernel with one
    traversal per ray and iso value with the per-pixel check counts (the reference's RayRes.checked_nodes) tapped, the pixel
    with the most checks is found on the device, and the image row that holds it is then rendered for that iso value alone --
    a launch with about one ray per wave of a few CUs, whose kernel time is the latency of that walk (plus the launch and the
    LDS fill).  No cut of the iso range and no scheduling gets a sweep's launch below this."""
    sp = C.c_void_p(stream.cuda_stream)
    d_rgb = torch.empty((1, H, W, 3), dtype=torch.uint8, device="cuda")
    d_chk = torch.empty((H, W), dtype=torch.int32, device="cuda")
    taps = capi.Taps(None, None, d_chk.data_ptr(), None)
    capi.check(L.vr_set_kernel(r.handle, capi.KERNEL_PIPELINE), "vr_set_kernel")
    best = torch.zeros(256, dtype=torch.int32, device="cuda")
    where = torch.zeros(256, dtype=torch.int64, device="cuda")
    with torch.cuda.stream(stream):
        for iso in range(I0, I0 + IN):
            capi.check(L.vr_render_device(r.handle, C.byref(view), iso, 1, 0, H, d_rgb.data_ptr(), None, C.byref(taps), sp), "vr_render_device")
            m, at = d_chk.view(-1).max(0)
            best[iso], where[iso] = m, at
    torch.cuda.synchronize()
    iso = int(best.argmax().item())
    checks, at = int(best[iso].item()), int(where[iso].item())
    y, x = divmod(at, W)
    capi.check(L.vr_set_timing(r.handle, 1), "vr_set_timing")
    tot, n = C.c_double(), C.c_uint32()
    times = []
    for i in range(12):
        with torch.cuda.stream(stream):
            capi.check(L.vr_render_device(r.handle, C.byref(view), iso, 1, y, y + 1, d_rgb.data_ptr(), None, None, sp), "vr_render_device")
        torch.cuda.synchronize()
        capi.check(L.vr_timing_read(r.handle, C.byref(tot), C.byref(n)), "vr_timing_read")
        if i >= 2:
            times.append(tot.value / max(1, n.value))
    capi.check(L.vr_set_timing(r.handle, 0), "vr_set_timing")
    return {"ms": min(times), "iso": iso, "pixel": [x, y], "checks": checks,
            "how": "kernel time of image row %d alone at iso %d with the one-traversal-per-iso kernel (min of 10)" % (y, iso)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                    help="auto: strong (ONE sweep split over the ranks), with the weak figure (one sweep per rank) as an extra")
    ap.add_argument("--shard", choices=["bands", "rows", "iso"], default="iso",
                    help="strong scaling: blocks of iso values (default), rows in bands of 32 dealt out round-robin, or contiguous row blocks")
    ap.add_argument("--gather", action="store_true", help="gather the framebuffers to rank 0 inside the timed region of the main figure")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: only the main figure, not the other scaling / the gather variant")
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--volume", choices=["skull", "c60", "standin256", "gyroid1024"], default="skull")
    ap.add_argument("--iso-begin", type=int, default=0)
    ap.add_argument("--iso-count", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-first-call", action="store_true", help="profiling runs: every launch of the run is a timed-step launch")
    ap.add_argument("--no-latency-floor", action="store_true", help="skip the search for the sweep's longest single walk (256 short launches)")
    ap.add_argument("--no-host-delivered", action="store_true",
                    help="skip the host-delivered leg (profiling runs: its shorter launches would mix into the kernel statistics)")
    ap.add_argument("--settle", type=int, default=24,
                    help="at most this many extra untimed launches before the warm-up, ended as soon as the library has "
                         "settled the sweep kernel's cut of the iso range for the workload (it tries a few per workload)")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="independent steps in flight per rank (own context, stream and frame buffer each): with 2, the "
                         "tail of one sweep, where its last long rays leave CUs idle, overlaps the start of the next; "
                         "the default 1 runs the steps strictly one after the other")
    args = ap.parse_args()

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
    # auto: ONE sweep split over the ranks by iso value (strong scaling: the reference's 0..255 sweep sharded over the GPUs of
    # a node); a sweep per rank (weak scaling: a stream of sweeps) and the other splits ride along as extras in the same line
    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    if args.shard == "iso" and args.iso_count < world:
        args.shard = "bands"  # fewer frames than ranks (one large frame): the rows are split instead
    if args.in_flight < 1 or (args.in_flight > 1 and (args.gather or world > 1)):
        raise SystemExit("--in-flight must be >= 1 and goes with one GPU and no gather")

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
    view = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    # A stream of our own: NULL would mean the context's private non-blocking stream, which torch's streams do not
    # order against -- the gather (and any copy torch makes) has to come behind the render on one stream.
    stream = torch.cuda.Stream()
    sp = C.c_void_p(stream.cuda_stream)

    class Plan:
        """What this rank renders in one step of one scaling mode, and how the parts go back together."""

        def __init__(self, mode, shard):
            self.mode, self.shard = mode, shard
            self.iso_begin, self.iso_count, self.bands = I0, IN, None
            self.row_begin, self.row_end = 0, H
            self.row_lists = None
            self.gatherer = None
            if mode == "strong" and world > 1:
                if shard == "iso":
                    self.iso_begin, self.iso_count = sharding.iso_shard(rank, world, I0, IN)
                    self.counts = [sharding.iso_shard(k, world, I0, IN)[1] for k in range(world)]
                elif shard == "rows":
                    self.row_begin, self.row_end = sharding.row_shard(rank, world, H)
                    self.row_lists = [list(range(*sharding.row_shard(k, world, H))) for k in range(world)]
                else:
                    self.bands = sharding.band_shard(rank, world, H)
                    self.row_lists = [sharding.band_rows_of(k, world, H) for k in range(world)]
            self.n_rows = self.bands["n_rows"] if self.bands else self.row_end - self.row_begin
            self.rays = self.iso_count * W * self.n_rows
            self.d_rgb = torch.empty((max(1, self.iso_count), max(1, self.n_rows), W, 3), dtype=torch.uint8, device="cuda")
            self.d_stats = torch.zeros((max(1, self.iso_count), 7), dtype=torch.int64, device="cuda")

        def render(self, rr, rgb, stats, s):
            if self.rays == 0:
                return
            if self.bands:
                b = self.bands
                capi.check(L.vr_render_device_bands(rr.handle, C.byref(view), self.iso_begin, self.iso_count, b["row_begin"],
                                                    b["band_rows"], b["band_pitch"], b["n_bands"], rgb.data_ptr(),
                                                    stats.data_ptr(), None, s), "vr_render_device_bands")
            else:
                capi.check(L.vr_render_device(rr.handle, C.byref(view), self.iso_begin, self.iso_count, self.row_begin,
                                              self.row_end, rgb.data_ptr(), stats.data_ptr(), None, s), "vr_render_device")

        def gather(self):
            """Framebuffers to rank 0: grouped isend / irecv straight into one preallocated [iso, H, W, 3] tensor (iso shards)
            or into one staging buffer per rank and one indexed copy (row shards); built once, allocates nothing per step."""
            local = self.d_rgb.cpu() if rehearsal else self.d_rgb
            if self.gatherer is None:
                if self.row_lists is not None:
                    self.gatherer = sharding.FrameGather(local, row_lists=self.row_lists, height=H, dst=0)
                elif self.mode == "strong" and self.shard == "iso":
                    self.gatherer = sharding.FrameGather(local, counts=self.counts, dst=0)
                else:
                    self.gatherer = sharding.FrameGather(local, counts=[IN] * world, dst=0)
            return self.gatherer(local)

        def describe(self):
            if world == 1:
                return "none"
            if self.mode == "weak":
                return "one sweep per rank"
            return {"iso": "iso-blocks", "rows": "row-blocks", "bands": "row-bands of 32, round-robin"}[self.shard]

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    settled = [0]

    def measure(plan, gather, slots=None):
        """W warmup steps, then exactly K timed steps between barriers; returns (max-over-ranks seconds, rays per step
        of the whole job, kernel ms per launch on this rank)."""
        slots = slots or [(r, plan.d_rgb, plan.d_stats, stream)]
        step_no = [0]

        def step():
            rr, rgb_k, stats_k, s_k = slots[step_no[0] % len(slots)]
            step_no[0] += 1
            with torch.cuda.stream(s_k):
                plan.render(rr, rgb_k, stats_k, C.c_void_p(s_k.cuda_stream))
                if gather and world > 1:
                    plan.gather()

        settle_steps = 0
        if plan.rays and len(slots) == 1:
            si = capi.SweepInfo()
            for _ in range(args.settle):  # untimed, before the warm-up: the library tries its cuts of the iso range
                rr, rgb_k, stats_k, s_k = slots[0]
                with torch.cuda.stream(s_k):  # the render alone: ranks settle after different numbers of launches,
                    plan.render(rr, rgb_k, stats_k, C.c_void_p(s_k.cuda_stream))  # so nothing collective may ride along
                torch.cuda.synchronize()
                settle_steps += 1
                if L.vr_sweep_info_read(r.handle, C.byref(si)) != capi.VR_OK or si.cut_settled:
                    break  # settled, or not a sweep-kernel workload: nothing to settle
        settled[0] = settle_steps
        for _ in range(args.warmup):
            step()
            torch.cuda.synchronize()  # untimed; lets the library see each warm-up launch's report before the next one
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
        if len(slots) > 1:
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
        rays = torch.tensor([float(plan.rays)], dtype=torch.float64, device=red_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        return float(t.item()), float(rays.item()), tot_ms.value / max(1, launches.value), int(launches.value)

    main_plan = Plan(scaling, args.shard)
    slots = [(r, main_plan.d_rgb, main_plan.d_stats, stream)]
    for _ in range(1, args.in_flight):  # further slots, each with its own context (scratch), stream and output buffers
        r2 = vr.Renderer(local_rank)
        r2.upload_volume(data)
        r2.build_kdt("AlwaysCenter")
        r2.set_kernel(args.kernel)
        slots.append((r2, torch.empty_like(main_plan.d_rgb), torch.zeros_like(main_plan.d_stats), torch.cuda.Stream()))
    dt, rays_per_step, kernel_ms, launches = measure(main_plan, args.gather, slots)
    main_settle = settled[0]

    stats = main_plan.d_stats.cpu().numpy()
    nodes_checked = int(stats[:, 0].sum()) if main_plan.rays else 0
    errors = torch.tensor([float(stats[:, 6].sum()) if main_plan.rays else 0.0], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(errors, op=dist.ReduceOp.SUM)
    sweep_kernel = args.kernel in (5, 6) or (args.kernel == 0 and main_plan.iso_count >= 8)  # kernels that report on themselves
    info = None
    if sweep_kernel and main_plan.rays:
        si = capi.SweepInfo()
        torch.cuda.synchronize()
        if L.vr_sweep_info_read(r.handle, C.byref(si)) == capi.VR_OK and si.tick_end > si.tick_begin:
            span = float(si.tick_end - si.tick_begin)
            info = {"walks_per_launch": int(si.walks), "items_per_launch": int(si.items),
                    "items_per_lane": int(si.items_per_lane), "cut_settled": bool(si.cut_settled), "settle_launches": main_settle,
                    "list_runs": int(si.list_runs), "tiles_longest_first": bool(si.tiles_longest_first),
                    "drain_frac": (si.tick_end - min(si.tick_queue_empty, si.tick_end)) / span,
                    "kernel_span_ms_gpu_clock": span / 1e5}

    extras = {}
    if world > 1 and not args.no_extras:
        other = "bands" if args.shard != "bands" else "iso"
        for name, mk, gather in (("weak_scaling", lambda: Plan("weak", args.shard), False),
                                 ("strong_scaling", lambda: Plan("strong", args.shard), False),
                                 ("strong_" + other, lambda: Plan("strong", other), False),
                                 ("strong_with_gather", lambda: Plan("strong", args.shard), True)):
            if (name == "weak_scaling" and scaling == "weak" and not args.gather) or \
               (name == "strong_scaling" and scaling == "strong" and not args.gather) or \
               (name == "strong_with_gather" and scaling == "strong" and args.gather):
                continue  # that is the main figure
            plan = mk()
            dt_x, rays_x, kms_x, _ = measure(plan, gather)
            extras[name] = {"value": rays_x * args.steps / dt_x / 1e6, "unit": "Mrays/s", "ms_per_step": dt_x / args.steps * 1e3,
                            "kernel_ms_rank0": kms_x, "sharding": plan.describe(), "gather": gather,
                            "rays_per_step": int(rays_x)}
            del plan

    if world == 1 and not args.no_extras and args.in_flight == 1 and sweep_kernel and not wide and main_plan.rays:
        # A stream of independent sweeps: two in flight (a second context with its own stream and frame buffer), so that one
        # sweep's memset and pre-pass and the start of its kernel fall into the other's drain.  Not the headline: `value` is one
        # sweep after the other on one stream.
        r2 = vr.Renderer(local_rank)
        r2.upload_volume(data)
        r2.build_kdt("AlwaysCenter")
        r2.set_kernel(args.kernel)
        two = [slots[0], (r2, torch.empty_like(main_plan.d_rgb), torch.zeros_like(main_plan.d_stats), torch.cuda.Stream())]
        si2 = capi.SweepInfo()
        for _ in range(args.settle):  # the second context settles its cut and learns its tile order like the first one did
            with torch.cuda.stream(two[1][3]):
                main_plan.render(r2, two[1][1], two[1][2], C.c_void_p(two[1][3].cuda_stream))
            torch.cuda.synchronize()
            if L.vr_sweep_info_read(r2.handle, C.byref(si2)) != capi.VR_OK or si2.cut_settled:
                break
        dt_x, rays_x, _, _ = measure(main_plan, False, two)
        extras["two_sweeps_in_flight"] = {"value": rays_x * args.steps / dt_x / 1e6, "unit": "Mrays/s", "ms_per_step": dt_x / args.steps * 1e3,
                                          "note": "the same step alternating between two contexts, each with its own stream and frames"}
        del two
        r2.close()

    if world == 1 and not args.no_extras and args.in_flight == 1 and sweep_kernel and main_plan.rays:
        # The same step for a caller that wants frames only (d_stats = NULL): the union kernel then runs with everything that only
        # serves the per-iso counters compiled out.  Not the headline: the reference's renderImage returns nodes_checked
        # (src/raytrace.zig:21,105), so `value` is measured with the counters.
        class _NoStats:
            @staticmethod
            def data_ptr():
                return None
        dt_n, rays_n, kms_n, _ = measure(main_plan, False, [(r, main_plan.d_rgb, _NoStats(), stream)])
        extras["without_counters"] = {"value": rays_n * args.steps / dt_n / 1e6, "unit": "Mrays/s", "ms_per_step": dt_n / args.steps * 1e3,
                                      "kernel_ms": kms_n, "note": "the same step with d_stats = NULL: frames only, no per-iso counters"}

    if rank == 0:
        vol_name = {"skull": "Skull.vol", "c60": "C60.vol", "standin256": "256^3 stand-in", "gyroid1024": "1024^3 gyroid"}[args.volume]
        workload = "%s_%dx%d_iso%d-%d_sweep" % (args.volume, W, H, I0, I0 + IN - 1)
        kernel_name = ("vr_k_render_wide" if args.kernel in (1, 2, 3) else "vr_k_sweep" if args.kernel == 5 else
                       "vr_k_union" if sweep_kernel else "vr_k_trace_wide") if wide else (
            {0: "vr_k_union" if main_plan.iso_count >= 8 else "vr_k_trace", 4: "vr_k_trace", 5: "vr_k_sweep", 6: "vr_k_union"}.get(args.kernel, "vr_k_render"))
        alg_bytes = 4.0 * nodes_checked + 3.0 * main_plan.rays
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        counters, traffic, stale = (None, None, None)
        if world == 1 and args.kernel == 0:
            counters, traffic, stale = profile_counters(workload, kernel_name, kernel_ms)
        out = {
            "metric": "Mrays/s on the %s iso sweep (iso %d..%d) at %dx%d" % (vol_name, I0, I0 + IN - 1, W, H),
            "value": rays_per_step * args.steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": what,
            "config": {"workload": workload,
                       "volume": "%s %dx%dx%d" % ((args.volume,) + tuple(data.resulution)),
                       "tree": "SlimKDT AlwaysCenter built on the GPU, %d nodes, depth %d" % (
                           build.number_of_nodes, build.layer_count),
                       "rays_per_step": int(rays_per_step),
                       "sharding": main_plan.describe(),
                       "gather": bool(args.gather and world > 1),
                       "kernel": "wide" if wide else capi.lib().vr_kernel_name(args.kernel).decode()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "hbm_traffic_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and kernel_ms > 0 else None,
                         "counters": counters, "counters_stale": stale,
                         "kernel": kernel_name, "kernel_ms": kernel_ms, "launches_timed": launches,
                         "algorithmic_bytes_per_launch": alg_bytes, "nodes_checked_per_launch": nodes_checked,
                         "measured_copy_GBps": copy_ceiling_gbps(torch),
                         "note": ("algorithmic bytes = the reference's node reads (4 B x nodes_checked) + 3 B per pixel, for "
                                  "rank 0's launch; the kernels read far fewer nodes (resume traversal; the sweep kernel also "
                                  "shares traversals between iso values), so frac prices the reference's work and can exceed "
                                  "1 -- hbm_traffic_frac is what HBM itself moves against its peak (counter pass of this "
                                  "workload; null when none is committed for it)")},
            "errors": int(errors.item()),
            "steps_in_flight": args.in_flight,
        }
        if info:
            out["roofline"].update(info)
        rf = out["roofline"]
        # What a step can be held against.  The frames are written once whatever the kernel does (3 B x pixels x iso values), and
        # the kernel's own reads come from HBM as far as the counter pass saw them: at the HBM peak that is the step's floor.
        frame_bytes = 3.0 * main_plan.rays
        own_reads = (counters or {}).get("hbm_read_bytes") or 0.0
        rf["hbm_floor_ms"] = (frame_bytes + own_reads) / (HBM_PEAK_GBS * 1e9) * 1e3
        rf["hbm_floor_bytes"] = {"frames_written_once": frame_bytes, "kernel_hbm_reads_counter_pass": own_reads or None}
        rf["step_over_hbm_floor"] = out["ms_per_step"] / rf["hbm_floor_ms"] if rf["hbm_floor_ms"] > 0 else None
        rf["infinity_cache"] = ("hit rate not measured: rocprofv3 exposes no Infinity Cache (MALL) counter on gfx950; FETCH_SIZE / "
                                "TCC_EA0_RDREQ count the L2's requests to the fabric, in front of that cache")
        if world == 1 and not wide and sweep_kernel and not args.no_latency_floor and args.in_flight == 1:
            lw = longest_walk(L, C, torch, capi, r, view, W, H, I0, IN, stream)
            r.set_kernel(args.kernel)
            rf["latency_floor_ms"] = lw["ms"]
            rf["latency_floor"] = lw
            rf["kernel_over_latency_floor"] = kernel_ms / lw["ms"] if lw["ms"] > 0 else None
            if lw["ms"] > rf["hbm_floor_ms"]:
                rf["bound"] = "latency"
                rf["bound_note"] = ("the launch cannot end before its longest single (ray, iso value) walk does, and that walk alone takes "
                                    "longer than moving the step's bytes at the HBM peak: `frac` (contract: algorithmic bytes / kernel "
                                    "time / 8 TB/s) stays a work ratio, hbm_traffic_frac is what HBM itself carries")
        out.update(extras)
        if rehearsal:
            out["rehearsal"] = "all ranks on one GPU over gloo: control-flow check only, not a measurement"
        if wide:
            out["build_ms_gpu"] = build_ms  # the host builder and the CPU oracle cannot run at this size
        if (world == 1 and info and (info.get("tiles_longest_first") or os.environ.get("VR_BENCH_FIRST_CALL")) and not main_plan.bands
                and args.in_flight == 1 and not args.no_first_call and IN >= 9):
            # The timed steps repeat one call, and a call the context remembers starts its tiles longest first, with the cut
            # of the iso range it has settled on.  What a caller gets who renders a view ONCE: a context that has never seen
            # the call (a fresh one per sample; its scratch is allocated by a call with one iso value less, which is another
            # workload and another call to the context), one launch, synchronised.
            def timed_call(rr):
                tot, n = C.c_double(), C.c_uint32()
                capi.check(L.vr_timing_read(rr.handle, C.byref(tot), C.byref(n)), "vr_timing_read")  # drop what came before
                t0 = time.perf_counter()
                with torch.cuda.stream(stream):
                    capi.check(L.vr_render_device(rr.handle, C.byref(view), I0, IN, 0, H, main_plan.d_rgb.data_ptr(),
                                                  main_plan.d_stats.data_ptr(), None, C.c_void_p(stream.cuda_stream)), "vr_render_device")
                torch.cuda.synchronize()
                dt1 = time.perf_counter() - t0
                capi.check(L.vr_timing_read(rr.handle, C.byref(tot), C.byref(n)), "vr_timing_read")
                return dt1 * 1e3, tot.value / max(1, n.value)

            def first_sample():
                with vr.Renderer(local_rank) as rr:
                    rr.upload_volume(data)
                    rr.build_kdt("AlwaysCenter")
                    rr.set_kernel(args.kernel)
                    with torch.cuda.stream(stream):
                        capi.check(L.vr_render_device(rr.handle, C.byref(view), I0, IN - 1, 0, H, main_plan.d_rgb.data_ptr(), None, None,
                                                      C.c_void_p(stream.cuda_stream)), "vr_render_device")
                    torch.cuda.synchronize()
                    capi.check(L.vr_set_timing(rr.handle, 1), "vr_set_timing")
                    return timed_call(rr)

            def other_then_repeat():
                with torch.cuda.stream(stream):  # a different call in between: the context remembers the timed steps' call across it
                    capi.check(L.vr_render_device(r.handle, C.byref(view), I0, min(8, IN), 0, min(8, H), main_plan.d_rgb.data_ptr(),
                                                  None, None, C.c_void_p(stream.cuda_stream)), "vr_render_device")
                torch.cuda.synchronize()
                return timed_call(r)

            firsts = sorted(first_sample() for _ in range(7))
            capi.check(L.vr_set_timing(r.handle, 1), "vr_set_timing")
            repeats = sorted(timed_call(r) for _ in range(7))
            others = sorted(other_then_repeat() for _ in range(7))
            capi.check(L.vr_set_timing(r.handle, 0), "vr_set_timing")
            out["first_call"] = {"ms_per_step": firsts[3][0], "kernel_ms": sorted(k for _, k in firsts)[3],
                                 "value": rays_per_step / (firsts[3][0] * 1e-3) / 1e6,
                                 "repeated_call_same_way_ms": repeats[3][0],
                                 "repeated_after_another_call_ms": others[3][0],
                                 "note": "medians of 7 single launches, each synchronised; first = the call on a context that has never "
                                         "seen it (tiles ordered by their rays' chords, base cut of the iso range), repeated = the timed "
                                         "steps' call again, after another call = the same with a different call on the context in between"}
        if world == 1 and not wide and not args.no_host_delivered and IN * W * H * 3 <= (1 << 30):
            # the sweep as the reference's caller gets it: frames in host memory (page-locked), src/main.zig:119-135
            try:
                buf = r.pinned_frames(IN, H, W)
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    r.renderSweep(pos, forw, up, W, H, 2.0, I0, IN, out=buf)
                    best = min(best, time.perf_counter() - t0)
                out["value_host_delivered"] = IN * W * H / best / 1e6
                out["host_delivered_ms_per_step"] = best * 1e3
                del buf
            except capi.VrError as e:
                out["value_host_delivered"] = None
                out["host_delivered_error"] = str(e)
        if not wide and not args.no_cpu_baseline and world == 1:
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

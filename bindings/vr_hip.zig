//! vr_hip.zig -- Zig side of the C ABI of libvr_hip.so (include/vr.h).
//!
//! UNTESTED: there is no zig toolchain in the build image, so this file has never been compiled.
//! It declares the entry points with `extern fn` and keeps the reference's call shape:
//!
//!   raytrace.renderImage(src, forw, up, img, viewport_width, kdt, isoval, dbp, inner_dbp)
//!       !struct { nodes_checked: u64 }                         (src/raytrace.zig:11-21)
//!
//! so that `main.zig` only has to swap its import.  Vec3(f32) is a packed
//! struct and *zigimg.Image a Zig type: neither crosses the boundary, plain floats and pointers do.
const std = @import("std");
const geo = @import("geo.zig");
const zigimg = @import("zigimg");
const SlimKDT = @import("SlimKDT.zig");
const ISOVAL = @import("typedef.zig").ISOVAL;

pub const Ctx = opaque {};

pub const View = extern struct {
    src: [3]f32,
    forw: [3]f32,
    up: [3]f32,
    viewport_width: f32,
    width: u32,
    height: u32,
};

pub const RenderStats = extern struct {
    nodes_checked: u64,
    hit_count: u64,
    miss_count: u64,
    oob_count: u64,
    escape_help_events: u64,
    moves: u64,
    errors: u64,
};

/// PartitionBuildData (src/SlimKDT.zig:215-219)
pub const BuildStats = extern struct { non_center_partition_count: u64, layer_count: u64, number_of_nodes: u64 };

pub extern fn vr_strerror(code: c_int) [*:0]const u8;
pub extern fn vr_ctx_create(device: c_int, out: *?*Ctx) c_int;
pub extern fn vr_ctx_destroy(ctx: ?*Ctx) void;
pub extern fn vr_upload_kdt(ctx: ?*Ctx, nodes: [*]const u32, n_slots: u64, root_dim: *const [3]u8) c_int;
pub extern fn vr_upload_volume(ctx: ?*Ctx, vox: [*]const u8, rx: u32, ry: u32, rz: u32) c_int;
pub extern fn vr_build_kdt(ctx: ?*Ctx, heuristic: c_int, stats_out: ?*BuildStats) c_int;
pub extern fn vr_download_kdt(ctx: ?*Ctx, nodes_out: ?[*]u32, capacity: u64, n_slots_out: *u64, root_dim_out: *[3]u8) c_int;
// volumes beyond SPACESIZE = u8 (no counterpart in the reference): wide node words, u16 root_dim
pub extern fn vr_host_check_kdt_wide(nodes: [*]const u32, n_slots: u64, root_dim: *const [3]u16, max_depth_out: ?*u32, n_nodes_out: ?*u64) c_int;
pub extern fn vr_upload_kdt_wide(ctx: ?*Ctx, nodes: [*]const u32, n_slots: u64, root_dim: *const [3]u16) c_int;
pub extern fn vr_download_kdt_wide(ctx: ?*Ctx, nodes_out: ?[*]u32, first_slot: u64, n_slots: u64, n_slots_out: ?*u64, root_dim_out: ?*[3]u16) c_int;
pub extern fn vr_save_kdt(path: [*:0]const u8, nodes: [*]const u32, n_slots: u64, root_dim: *const [3]u8) c_int;
pub extern fn vr_load_kdt(path: [*:0]const u8, nodes_out: *?[*]u32, n_slots_out: *u64, root_dim_out: *[3]u8) c_int;
// one host thread, one context per GPU: shard = 0 splits the iso values, 1 the rows of every frame
pub extern fn vr_mgpu_render_sweep(ctxs: [*]const ?*Ctx, n_ctx: u32, view: *const View, iso_begin: u32, iso_count: u32, shard: c_int, rgb24_out: [*]u8, stats_out: ?[*]RenderStats) c_int;
pub extern fn vr_free(p: ?*anyopaque) void;
pub extern fn vr_render(ctx: ?*Ctx, view: *const View, isoval: u8, rgb24_out: [*]u8, stats_out: ?*RenderStats, taps: ?*const anyopaque) c_int;
/// rows [row_begin,row_end) of iso_count frames into DEVICE memory, asynchronous; `stream` is a hipStream_t
/// (null = the context's own non-blocking stream: wait for it with vr_sync)
pub extern fn vr_render_device(ctx: ?*Ctx, view: *const View, iso_begin: u32, iso_count: u32, row_begin: u32, row_end: u32, d_rgb24: ?*anyopaque, d_stats: ?*anyopaque, d_taps: ?*const anyopaque, stream: ?*anyopaque) c_int;
/// the same call with its scratch in a static device array: no allocation inside, at most 2^20 pixels per call
pub extern fn vr_render_device_noalloc(ctx: ?*Ctx, view: *const View, iso_begin: u32, iso_count: u32, row_begin: u32, row_end: u32, d_rgb24: ?*anyopaque, d_stats: ?*anyopaque, d_taps: ?*const anyopaque, stream: ?*anyopaque) c_int;
pub extern fn vr_sync(ctx: ?*Ctx) c_int;
/// page-locked frame buffers for vr_render / vr_render_sweep (full PCIe rate); release with vr_free_host
pub extern fn vr_alloc_host(bytes: usize, out: *?*anyopaque) c_int;
pub extern fn vr_free_host(p: ?*anyopaque) void;
/// SlimKDT.newHeursiticPartition on the host (src/SlimKDT.zig:222-322); release nodes_out with vr_free
pub extern fn vr_host_build_kdt(vox: [*]const u8, res: *const [3]u32, heuristic: c_int, max_slots: u64, nodes_out: *?[*]u32, n_slots_out: *u64, root_dim_out: *[3]u8, stats_out: ?*BuildStats) c_int;
pub extern fn vr_render_sweep(ctx: ?*Ctx, view: *const View, iso_begin: u32, iso_count: u32, rgb24_out: [*]u8, stats_out: ?[*]RenderStats, taps: ?*const anyopaque) c_int;

pub const Error = error{GpuRenderFailed};

/// Uploads kdt.nodes verbatim.  SlimNode is a packed struct of 26 bits in a 4-byte backing integer
/// (src/SlimKDT.zig:9-13), so []SlimNode is already the dense little-endian u32 array the library reads.
pub fn uploadKdt(ctx: *Ctx, kdt: *const SlimKDT) Error!void {
    const words: [*]const u32 = @ptrCast(@alignCast(kdt.nodes.ptr));
    if (vr_upload_kdt(ctx, words, kdt.nodes.len, &kdt.root_dim) != 0) return Error.GpuRenderFailed;
}

/// Drop-in for raytrace.renderImage: same arguments plus the context that holds the uploaded tree.
pub fn renderImage(
    ctx: *Ctx,
    src: geo.VecF,
    forw: geo.VecF,
    up: geo.VecF,
    img: *zigimg.Image,
    viewport_width: f32,
    isoval: ISOVAL,
) Error!struct { nodes_checked: u64 } {
    const view = View{
        .src = .{ src.x, src.y, src.z },
        .forw = .{ forw.x, forw.y, forw.z },
        .up = .{ up.x, up.y, up.z },
        .viewport_width = viewport_width,
        .width = @intCast(img.width),
        .height = @intCast(img.height),
    };
    var stats: RenderStats = undefined;
    // zigimg's rgb24 pixel storage is a dense []Rgb24 (3 bytes per pixel, row-major, top row first)
    const out: [*]u8 = @ptrCast(img.pixels.rgb24.ptr);
    if (vr_render(ctx, &view, isoval, out, &stats, null) != 0) return Error.GpuRenderFailed;
    return .{ .nodes_checked = stats.nodes_checked };
}

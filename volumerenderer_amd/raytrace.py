"""raytrace: renderImage and the iso sweep on the GPU (src/raytrace.zig:11-106, src/main.zig:111-138)."""
import ctypes as C

import numpy as np

from . import capi


def main_camera(resulution, pos=(-30.0, -300.0, -30.0)):
    """Camera of the reference driver (src/main.zig:31-43): returns (pos, forw, up) float32 arrays."""
    L = capi.lib()
    r = (C.c_uint32 * 3)(*[int(x) for x in resulution])
    p = (C.c_float * 3)(*[float(x) for x in pos])
    f = (C.c_float * 3)()
    u = (C.c_float * 3)()
    L.vr_main_camera(r, p, f, u)
    return (np.array(list(p), dtype=np.float32), np.array(list(f), dtype=np.float32),
            np.array(list(u), dtype=np.float32))


class Image:
    """Stand-in for zigimg.Image with .rgb24 pixels: width, height, pixels[height, width, 3] uint8."""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self.pixels = np.zeros((self.height, self.width, 3), dtype=np.uint8)


def _view(src, forw, up, width, height, viewport_width):
    v = capi.View()
    for k in range(3):
        v.src[k] = float(src[k])
        v.forw[k] = float(forw[k])
        v.up[k] = float(up[k])
    v.viewport_width = float(viewport_width)
    v.width, v.height = int(width), int(height)
    return v


class Renderer:
    """One device context (vr_ctx) holding an uploaded SlimKDT."""

    def __init__(self, device=0):
        self._L = capi.lib()
        self._ctx = C.c_void_p()
        capi.check(self._L.vr_ctx_create(int(device), C.byref(self._ctx)), "vr_ctx_create")
        self.kdt = None

    def close(self):
        if self._ctx:
            self._L.vr_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._ctx

    def set_kernel(self, variant):
        capi.check(self._L.vr_set_kernel(self._ctx, int(variant)), "vr_set_kernel")

    def upload_volume(self, data):
        rx, ry, rz = data.resulution
        capi.check(self._L.vr_upload_volume(self._ctx, data.data.ctypes.data, rx, ry, rz), "vr_upload_volume")

    def upload_kdt(self, kdt):
        if getattr(kdt, "wide", False):
            rd = (C.c_uint16 * 3)(*[int(x) for x in kdt.root_dim])
            capi.check(self._L.vr_upload_kdt_wide(self._ctx, kdt.nodes.ctypes.data, len(kdt.nodes), rd),
                       "vr_upload_kdt_wide")
        else:
            rd = (C.c_uint8 * 3)(*[int(x) for x in kdt.root_dim])
            capi.check(self._L.vr_upload_kdt(self._ctx, kdt.nodes.ctypes.data, len(kdt.nodes), rd), "vr_upload_kdt")
        self.kdt = kdt

    def build_kdt(self, heuristic="AlwaysCenter"):
        h = capi.HEURISTICS[heuristic] if isinstance(heuristic, str) else int(heuristic)
        st = capi.BuildStats()
        capi.check(self._L.vr_build_kdt(self._ctx, h, C.byref(st)), "vr_build_kdt")
        return st

    def download_kdt(self):
        n = C.c_uint64()
        rd = (C.c_uint8 * 3)()
        capi.check(self._L.vr_download_kdt(self._ctx, None, 0, C.byref(n), rd), "vr_download_kdt")
        nodes = np.zeros(n.value, dtype=np.uint32)
        capi.check(self._L.vr_download_kdt(self._ctx, nodes.ctypes.data, n.value, C.byref(n), rd), "vr_download_kdt")
        return nodes, np.array(list(rd), dtype=np.uint8)

    def download_kdt_wide(self, first_slot=0, n_slots=None):
        """Slots [first_slot, first_slot + n_slots) of a wide tree (a volume beyond 256 voxels per axis):
        returns (wide node words, root_dim uint16[3], array length on the device)."""
        n = C.c_uint64()
        rd = (C.c_uint16 * 3)()
        capi.check(self._L.vr_download_kdt_wide(self._ctx, None, 0, 0, C.byref(n), rd), "vr_download_kdt_wide")
        count = n.value - first_slot if n_slots is None else int(n_slots)
        nodes = np.zeros(count, dtype=np.uint32)
        capi.check(self._L.vr_download_kdt_wide(self._ctx, nodes.ctypes.data, int(first_slot), count, C.byref(n), rd),
                   "vr_download_kdt_wide")
        return nodes, np.array(list(rd), dtype=np.uint16), n.value

    def pinned_frames(self, iso_count, height, width):
        """uint8[iso_count, height, width, 3] in page-locked host memory (vr_alloc_host).  The allocation belongs
        to the returned array, not to the Renderer: it is released (vr_free_host) when the last view of the array
        has been collected, so frames stay readable after close()."""
        import weakref
        n = int(iso_count) * int(height) * int(width) * 3
        p = C.c_void_p()
        capi.check(self._L.vr_alloc_host(n, C.byref(p)), "vr_alloc_host")
        buf = (C.c_uint8 * n).from_address(p.value)  # every numpy view keeps `buf` alive through .base
        weakref.finalize(buf, self._L.vr_free_host, C.c_void_p(p.value))
        return np.frombuffer(buf, dtype=np.uint8).reshape(iso_count, height, width, 3)

    def renderSweep(self, src, forw, up, width, height, viewport_width, iso_begin, iso_count, taps=False, out=None):
        """iso loop of src/main.zig:111-138.  Returns dict(rgb[iso,h,w,3], stats[list], + taps).  `out`: frames
        buffer to fill (e.g. from pinned_frames) instead of a fresh pageable array."""
        v = _view(src, forw, up, width, height, viewport_width)
        rgb = out if out is not None else np.zeros((iso_count, height, width, 3), dtype=np.uint8)
        assert rgb.shape == (iso_count, height, width, 3) and rgb.dtype == np.uint8 and rgb.flags.c_contiguous
        stats = (capi.RenderStats * iso_count)()
        out = {"rgb": rgb}
        tp = None
        if taps:
            out["cell_flag"] = np.zeros((iso_count, height, width), dtype=np.uint32)
            out["t"] = np.zeros((iso_count, height, width), dtype=np.float32)
            out["checked"] = np.zeros((iso_count, height, width), dtype=np.uint32)
            out["node"] = np.zeros((iso_count, height, width), dtype=np.uint32)
            tp = capi.Taps(out["cell_flag"].ctypes.data, out["t"].ctypes.data, out["checked"].ctypes.data,
                           out["node"].ctypes.data)
        capi.check(self._L.vr_render_sweep(self._ctx, C.byref(v), int(iso_begin), int(iso_count), rgb.ctypes.data,
                                           stats, C.byref(tp) if tp else None), "vr_render_sweep")
        out["stats"] = [stats[i].as_dict() for i in range(iso_count)]
        return out

    def renderImage(self, src, forw, up, img, viewport_width, isoval, taps=False):
        """raytrace.renderImage (src/raytrace.zig:11-21): fills img.pixels, returns dict with nodes_checked."""
        out = self.renderSweep(src, forw, up, img.width, img.height, viewport_width, int(isoval), 1, taps=taps)
        img.pixels[...] = out["rgb"][0]
        res = {"nodes_checked": out["stats"][0]["nodes_checked"], "stats": out["stats"][0]}
        if taps:
            for k in ("cell_flag", "t", "checked", "node"):
                res[k] = out[k][0]
        return res


def renderSweepMulti(renderers, src, forw, up, width, height, viewport_width, iso_begin, iso_count, shard="iso", out=None):
    """The iso loop split over several Renderers (normally one per GPU, the same tree in each) by ONE host thread:
    vr_mgpu_render_sweep.  shard = "iso" (blocks of whole frames) or "rows" (a block of rows of every frame).
    Returns dict(rgb[iso,h,w,3], stats[list]) exactly as renderSweep on one Renderer does."""
    L = capi.lib()
    v = _view(src, forw, up, width, height, viewport_width)
    rgb = out if out is not None else np.zeros((iso_count, height, width, 3), dtype=np.uint8)
    assert rgb.shape == (iso_count, height, width, 3) and rgb.dtype == np.uint8 and rgb.flags.c_contiguous
    stats = (capi.RenderStats * iso_count)()
    handles = (C.c_void_p * len(renderers))(*[r.handle for r in renderers])
    capi.check(L.vr_mgpu_render_sweep(handles, len(renderers), C.byref(v), int(iso_begin), int(iso_count),
                                      capi.SHARD_ROWS if shard == "rows" else capi.SHARD_ISO, rgb.ctypes.data, stats),
               "vr_mgpu_render_sweep")
    return {"rgb": rgb, "stats": [stats[i].as_dict() for i in range(iso_count)]}


def renderImage(src, forw, up, img, viewport_width, kdt, isoval, device=0):
    """One-shot form with the reference's argument order (src/raytrace.zig:11-21)."""
    with Renderer(device) as r:
        r.upload_kdt(kdt)
        return r.renderImage(src, forw, up, img, viewport_width, isoval)

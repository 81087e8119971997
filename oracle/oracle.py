"""ctypes binding of the CPU oracle (oracle/vr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, the smoke check and bench.py's cpu_baseline leg as
the checker.  Nothing under volumerenderer_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvr_oracle.so")
# the same C file built with u16 coordinates: volumes the reference cannot represent (up to 1024 per axis)
_LIB_PATH_WIDE = os.path.join(_HERE, "libvr_oracle_wide.so")

HEURISTICS = {
    "AlwaysCenter": 0,
    "MinimizeSpan": 1,
    "FlexibleFirst5": 2,
    "FlexibleFirst10": 3,
    "FlexibleAfter10": 4,
    "FlexibleAfter15": 5,
    "FlexibleAfter20": 6,
    "FlexibleCenterWeighted5": 7,
    "FlexibleCenterWeighted10": 8,
    "FlexibleCenterWeighted30": 9,
    "FlexibleCenterWeighted50": 10,
    "FlexibleCenterWeighted75": 11,
    "FlexibleCenterWeighted100": 12,
}

FLAG_MISS, FLAG_HIT, FLAG_OOB = 0, 1, 2


class BuildStats(C.Structure):
    _fields_ = [("non_center_partition_count", C.c_uint64), ("layer_count", C.c_uint64),
                ("number_of_nodes", C.c_uint64)]


class RenderStats(C.Structure):
    _fields_ = [("nodes_checked", C.c_uint64), ("hit_count", C.c_uint64), ("miss_count", C.c_uint64),
                ("oob_count", C.c_uint64), ("escape_help_events", C.c_uint64), ("moves", C.c_uint64),
                ("max_checked_per_ray", C.c_uint64)]


def build(force=False):
    """Compile the C restatement with the recipe in oracle/Makefile."""
    src_time = os.path.getmtime(os.path.join(_HERE, "vr_oracle.c"))
    if force or any(not os.path.exists(q) or os.path.getmtime(q) < src_time for q in (_LIB_PATH, _LIB_PATH_WIDE)):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_libs = {}


def _coord(wide):
    return C.c_uint16 if wide else C.c_uint8


def lib(wide=False):
    if wide not in _libs:
        path = _LIB_PATH_WIDE if wide else _LIB_PATH
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u8p, u32p, f32p = C.POINTER(_coord(wide)), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
        L.vro_build_kdt.argtypes = [C.c_void_p, u32p, C.c_int, C.c_uint64, C.POINTER(u32p),
                                    C.POINTER(C.c_uint64), u8p, C.POINTER(BuildStats)]
        L.vro_build_kdt.restype = C.c_int
        L.vro_free.argtypes = [C.c_void_p]
        L.vro_free.restype = None
        L.vro_render_rows.argtypes = [C.c_void_p, C.c_uint64, u8p, f32p, f32p, f32p, C.c_uint32, C.c_uint32,
                                      C.c_float, C.c_uint8, C.c_uint32, C.c_uint32, C.c_void_p,
                                      C.POINTER(RenderStats), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vro_render_rows.restype = C.c_int
        L.vro_main_camera.argtypes = [u32p, f32p, f32p, f32p]
        L.vro_main_camera.restype = None
        L.vro_load_ucd.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), u32p]
        L.vro_load_ucd.restype = C.c_int
        _libs[wide] = L
    return _libs[wide]


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def load_ucd(path):
    """Data.loadUCD: returns (vox[x,y,z] uint8 array, (rx,ry,rz))."""
    L = lib()
    p = C.POINTER(C.c_uint8)()
    res = (C.c_uint32 * 3)()
    rc = L.vro_load_ucd(os.fsencode(path), C.byref(p), res)
    if rc != 0:
        raise RuntimeError("vro_load_ucd(%s) failed: %d" % (path, rc))
    n = res[0] * res[1] * res[2]
    arr = np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(res[0], res[1], res[2])
    L.vro_free(p)
    return arr


def main_camera(res, pos=(-30.0, -300.0, -30.0)):
    """Camera of the reference driver: returns (pos, forw, up) as float32 arrays."""
    L = lib()
    r = (C.c_uint32 * 3)(*[int(x) for x in res])
    f = (C.c_float * 3)()
    u = (C.c_float * 3)()
    L.vro_main_camera(r, _f3(pos), f, u)
    return (np.array(pos, dtype=np.float32), np.array(list(f), dtype=np.float32),
            np.array(list(u), dtype=np.float32))


def build_kdt(vox, heuristic="AlwaysCenter", max_slots=0, wide=False):
    """SlimKDT.newHeursiticPartition: returns (nodes uint32[n_slots], root_dim uint8[3], BuildStats).
    wide=True: u16 coordinates and the wide node word (root_dim comes back as uint16)."""
    L = lib(wide)
    vox = np.ascontiguousarray(vox, dtype=np.uint8)
    assert vox.ndim == 3
    res = (C.c_uint32 * 3)(*vox.shape)
    h = HEURISTICS[heuristic] if isinstance(heuristic, str) else int(heuristic)
    p = C.POINTER(C.c_uint32)()
    n = C.c_uint64()
    rd = (_coord(wide) * 3)()
    st = BuildStats()
    rc = L.vro_build_kdt(vox.ctypes.data, res, h, max_slots, C.byref(p), C.byref(n), rd, C.byref(st))
    if rc != 0:
        raise RuntimeError("vro_build_kdt failed: %d" % rc)
    nodes = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    L.vro_free(p)
    return nodes, np.array(list(rd), dtype=np.uint16 if wide else np.uint8), st


def trim_nodes(nodes):
    """Shrink a heap-indexed node array to the smallest power of two that holds every used slot."""
    nz = np.flatnonzero(nodes)
    last = int(nz[-1]) if len(nz) else 0
    n = 1
    while n <= last:
        n *= 2
    return np.ascontiguousarray(nodes[:n]) if n <= len(nodes) else nodes


def render(nodes, root_dim, src, forw, up, width, height, viewport_width, iso, rows=None, taps=True,
           out=None, wide=False):
    """raytrace.renderImage.  Returns dict(rgb, stats, cell_flag, t, checked, node).
    wide=True: nodes are wide node words; cell_flag packs x | y<<10 | z<<20 | flag<<30."""
    L = lib(wide)
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    rd = (_coord(wide) * 3)(*[int(x) for x in root_dim])
    rgb = out if out is not None else np.zeros((height, width, 3), dtype=np.uint8)
    r0, r1 = (0, height) if rows is None else rows
    st = RenderStats()
    if taps:
        cf = np.zeros((height, width), dtype=np.uint32)
        tt = np.zeros((height, width), dtype=np.float32)
        ck = np.zeros((height, width), dtype=np.uint32)
        nd = np.zeros((height, width), dtype=np.uint32)
        ptrs = [cf.ctypes.data, tt.ctypes.data, ck.ctypes.data, nd.ctypes.data]
    else:
        cf = tt = ck = nd = None
        ptrs = [None, None, None, None]
    rc = L.vro_render_rows(nodes.ctypes.data, len(nodes), rd, _f3(src), _f3(forw), _f3(up), width, height,
                           float(viewport_width), int(iso), r0, r1, rgb.ctypes.data, C.byref(st), *ptrs)
    if rc != 0:
        raise RuntimeError("vro_render_rows failed: %d" % rc)
    return {"rgb": rgb, "stats": st, "cell_flag": cf, "t": tt, "checked": ck, "node": nd}


def trilerp_batch(points):
    """trilerp (src/raytrace.zig:257-291) on an [n, 17] float32 array of (x, y, z, x0, x1, y0, y1, z0, z1, values[8])."""
    L = lib()
    pts = np.ascontiguousarray(points, dtype=np.float32)
    assert pts.ndim == 2 and pts.shape[1] == 17
    out = np.zeros(len(pts), dtype=np.float32)
    L.vro_trilerp_batch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.vro_trilerp_batch.restype = None
    L.vro_trilerp_batch(pts.ctypes.data, len(pts), out.ctypes.data)
    return out


def refine(vox, src, forw, up, width, height, viewport_width, cell_flag, t, rows=None, wide=False):
    """Density trilerp gives at every hit point inside its hit cell (0 where the ray did not hit); parity unpinned
    beyond trilerp and getCornerDens (src/Data.zig:102-114)."""
    L = lib(wide)
    vox = np.ascontiguousarray(vox, dtype=np.uint8)
    res = (C.c_uint32 * 3)(*vox.shape)
    cf = np.ascontiguousarray(cell_flag, dtype=np.uint32)
    tt = np.ascontiguousarray(t, dtype=np.float32)
    out = np.zeros((height, width), dtype=np.float32)
    r0, r1 = (0, height) if rows is None else rows
    L.vro_refine_rows.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                  C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.vro_refine_rows(vox.ctypes.data, res, _f3(src), _f3(forw), _f3(up), width, height, float(viewport_width), r0, r1,
                           cf.ctypes.data, tt.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("vro_refine_rows failed: %d" % rc)
    return out


class RayStep(C.Structure):
    _fields_ = [("t", C.c_float), ("p", C.c_float * 3), ("node", C.c_uint32), ("size", C.c_uint32), ("min", C.c_uint8),
                ("max", C.c_uint8), ("isoval", C.c_uint8), ("action", C.c_uint8)]


class RayReport(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3), ("t_entry", C.c_float), ("t_final", C.c_float),
                ("flag", C.c_uint32), ("n_steps", C.c_uint32), ("n_stored", C.c_uint32)]


def trace_ray(nodes, root_dim, src, forw, up, width, height, viewport_width, iso, xi, yi, cap=1 << 16):
    """traceRay of one pixel with every loop iteration recorded (the reference's dbp print as data): (report, steps)."""
    L = lib()
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    rd = (C.c_uint8 * 3)(*[int(x) for x in root_dim])
    steps = (RayStep * cap)()
    rep = RayReport()
    L.vro_trace_ray.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                C.c_uint32, C.c_uint32, C.c_float, C.c_uint8, C.c_uint32, C.c_uint32, C.POINTER(RayStep),
                                C.c_uint32, C.POINTER(RayReport)]
    rc = L.vro_trace_ray(nodes.ctypes.data, rd, _f3(src), _f3(forw), _f3(up), width, height, float(viewport_width), int(iso),
                         int(xi), int(yi), steps, cap, C.byref(rep))
    if rc != 0:
        raise RuntimeError("vro_trace_ray failed: %d" % rc)
    return rep, np.frombuffer(steps, dtype=np.uint8).reshape(cap, C.sizeof(RayStep))[: rep.n_stored].copy()

"""ctypes view of libvr_hip.so (C ABI declared in include/vr.h).

The library is the product; this module only loads it and declares the prototypes.  Loading fails
loudly when the shared object has not been built -- there is no Python or CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VR_LIB_PATH") or os.path.join(_HERE, "libvr_hip.so")  # VR_LIB_PATH: tuning builds

VR_OK = 0
VR_EARG, VR_ENOMEM, VR_ESTATE, VR_ETOODEEP, VR_EIO, VR_EFORMAT, VR_ENODEVICE = -1, -2, -3, -4, -5, -6, -7
VR_ETREE = -8
VR_ELIMIT = -9
VR_EHIP_BASE = -1000
FLAG_MISS, FLAG_HIT, FLAG_OOB = 0, 1, 2

SHARD_ISO, SHARD_ROWS = 0, 1
KERNEL_DEFAULT, KERNEL_RESTART_GLOBAL, KERNEL_RESTART_LDS, KERNEL_RESUME, KERNEL_PIPELINE, KERNEL_SWEEP = 0, 1, 2, 3, 4, 5
KERNEL_UNION = 6  # one traversal per ray for the whole iso range

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

# every symbol include/vr.h declares; tests check that the library exports each of them
EXPORTS = [
    "vr_strerror", "vr_load_ucd", "vr_save_ucd", "vr_save_kdt", "vr_load_kdt", "vr_host_build_kdt", "vr_host_prepare_kdt", "vr_host_check_kdt_wide", "vr_prepared_words", "vr_main_camera", "vr_free", "vr_alloc_host", "vr_free_host", "vr_write_png_rgb24", "vr_write_ppm_rgb24", "vr_ctx_create",
    "vr_ctx_destroy", "vr_upload_volume", "vr_upload_kdt", "vr_build_kdt", "vr_download_kdt", "vr_upload_kdt_wide", "vr_download_kdt_wide", "vr_render",
    "vr_render_sweep", "vr_render_device", "vr_render_device_noalloc", "vr_render_device_bands", "vr_mgpu_render_sweep", "vr_selftest", "vr_trace_ray", "vr_print_ray_trace", "vr_trilerp_device", "vr_refine_device", "vr_sync", "vr_sweep_info_read", "vr_set_timing", "vr_timing_read", "vr_set_kernel",
    "vr_kernel_name",
]


class BuildStats(C.Structure):
    _fields_ = [("non_center_partition_count", C.c_uint64), ("layer_count", C.c_uint64),
                ("number_of_nodes", C.c_uint64)]


class RenderStats(C.Structure):
    _fields_ = [("nodes_checked", C.c_uint64), ("hit_count", C.c_uint64), ("miss_count", C.c_uint64),
                ("oob_count", C.c_uint64), ("escape_help_events", C.c_uint64), ("moves", C.c_uint64),
                ("errors", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class View(C.Structure):
    _fields_ = [("src", C.c_float * 3), ("forw", C.c_float * 3), ("up", C.c_float * 3),
                ("viewport_width", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32)]


class SweepInfo(C.Structure):
    _fields_ = [("walks", C.c_uint64), ("items", C.c_uint64), ("tick_begin", C.c_uint64),
                ("tick_queue_empty", C.c_uint64), ("tick_end", C.c_uint64), ("items_per_lane", C.c_uint32),
                ("cut_settled", C.c_uint32), ("list_runs", C.c_uint32), ("tiles_longest_first", C.c_uint32),
                ("rendered_again", C.c_uint32), ("reserved", C.c_uint32)]


class RayStep(C.Structure):
    _fields_ = [("t", C.c_float), ("p", C.c_float * 3), ("node", C.c_uint32), ("size", C.c_uint32), ("min", C.c_uint8),
                ("max", C.c_uint8), ("isoval", C.c_uint8), ("action", C.c_uint8)]


class RayReport(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3), ("t_entry", C.c_float), ("t_final", C.c_float),
                ("flag", C.c_uint32), ("n_steps", C.c_uint32), ("n_stored", C.c_uint32)]


class Taps(C.Structure):
    _fields_ = [("cell_flag", C.c_void_p), ("t", C.c_void_p), ("checked", C.c_void_p), ("node", C.c_void_p)]


class VrError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__("%s failed: %d (%s)" % (what, code, strerror(code)))


_lib = None


def lib():
    """Loads libvr_hip.so; raises if it is missing (build it with __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libvr_hip.so is not built (%s): run `make -C volumerenderer_amd/csrc`; "
                           "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    u8p, u32p, f32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_void_p
    L.vr_strerror.argtypes = [C.c_int]
    L.vr_strerror.restype = C.c_char_p
    L.vr_load_ucd.argtypes = [C.c_char_p, C.POINTER(u8p), u32p]
    L.vr_save_ucd.argtypes = [C.c_char_p, vp, u32p]
    L.vr_save_kdt.argtypes = [C.c_char_p, vp, C.c_uint64, u8p]
    L.vr_load_kdt.argtypes = [C.c_char_p, C.POINTER(u32p), C.POINTER(C.c_uint64), u8p]
    L.vr_host_build_kdt.argtypes = [vp, u32p, C.c_int, C.c_uint64, C.POINTER(u32p), C.POINTER(C.c_uint64), u8p,
                                    C.POINTER(BuildStats)]
    L.vr_host_prepare_kdt.argtypes = [vp, C.c_uint64, u8p, vp, C.c_uint64, u32p, C.POINTER(C.c_uint64)]
    L.vr_host_check_kdt_wide.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint16), u32p, C.POINTER(C.c_uint64)]
    L.vr_prepared_words.argtypes = [C.c_uint64]
    L.vr_prepared_words.restype = C.c_uint64
    L.vr_main_camera.argtypes = [u32p, f32p, f32p, f32p]
    L.vr_main_camera.restype = None
    L.vr_write_png_rgb24.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    L.vr_write_ppm_rgb24.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    L.vr_free.argtypes = [vp]
    L.vr_free.restype = None
    L.vr_alloc_host.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.vr_free_host.argtypes = [vp]
    L.vr_free_host.restype = None
    L.vr_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.vr_ctx_destroy.argtypes = [vp]
    L.vr_ctx_destroy.restype = None
    L.vr_upload_volume.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.vr_upload_kdt.argtypes = [vp, vp, C.c_uint64, u8p]
    L.vr_build_kdt.argtypes = [vp, C.c_int, C.POINTER(BuildStats)]
    L.vr_download_kdt.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64), u8p]
    u16p = C.POINTER(C.c_uint16)
    L.vr_upload_kdt_wide.argtypes = [vp, vp, C.c_uint64, u16p]
    L.vr_download_kdt_wide.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), u16p]
    L.vr_render.argtypes = [vp, C.POINTER(View), C.c_uint8, vp, C.POINTER(RenderStats), C.POINTER(Taps)]
    L.vr_render_sweep.argtypes = [vp, C.POINTER(View), C.c_uint32, C.c_uint32, vp, C.POINTER(RenderStats),
                                  C.POINTER(Taps)]
    L.vr_render_device.argtypes = [vp, C.POINTER(View), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp,
                                   C.POINTER(Taps), vp]
    L.vr_render_device_noalloc.argtypes = L.vr_render_device.argtypes
    L.vr_render_device_bands.argtypes = [vp, C.POINTER(View), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, vp, vp, C.POINTER(Taps), vp]
    L.vr_mgpu_render_sweep.argtypes = [C.POINTER(vp), C.c_uint32, C.POINTER(View), C.c_uint32, C.c_uint32, C.c_int, vp,
                                       C.POINTER(RenderStats)]
    L.vr_selftest.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.vr_trace_ray.argtypes = [vp, C.POINTER(View), C.c_uint8, C.c_uint32, C.c_uint32, C.POINTER(RayStep), C.c_uint32,
                               C.POINTER(RayReport)]
    L.vr_print_ray_trace.argtypes = [C.POINTER(RayReport), C.POINTER(RayStep), C.c_char_p]
    L.vr_trilerp_device.argtypes = [vp, vp, C.c_uint64, vp, vp]
    L.vr_refine_device.argtypes = [vp, C.POINTER(View), C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
    L.vr_sync.argtypes = [vp]
    L.vr_sweep_info_read.argtypes = [vp, C.POINTER(SweepInfo)]
    L.vr_set_timing.argtypes = [vp, C.c_int]
    L.vr_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.vr_set_kernel.argtypes = [vp, C.c_int]
    L.vr_kernel_name.argtypes = [C.c_int]
    L.vr_kernel_name.restype = C.c_char_p
    _lib = L
    return L


def strerror(code):
    return lib().vr_strerror(int(code)).decode()


def check(code, what):
    if code != VR_OK:
        raise VrError(code, what)

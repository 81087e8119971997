"""MI355X-native isosurface ray casting: the hot path of VolumeRenderer behind its own call boundary.

Host-side mirror of the reference interface (thin ctypes wrappers over libvr_hip.so):

    Data.loadUCD(path)                                  src/Data.zig:9-44
    SlimKDT.newHeursiticPartition(data, heuristic)      src/SlimKDT.zig:222-246
    raytrace.renderImage(src, forw, up, img, viewport_width, kdt, isoval)   src/raytrace.zig:11-106
    raytrace.renderSweep(...)                           the iso loop of src/main.zig:111-138
"""
from . import capi  # noqa: F401
from .data import Data  # noqa: F401
from .slimkdt import SlimKDT  # noqa: F401
from . import raytrace  # noqa: F401
from .raytrace import Image, Renderer, main_camera  # noqa: F401
from . import datasets, sharding  # noqa: F401,E402

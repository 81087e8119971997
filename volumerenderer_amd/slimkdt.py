"""SlimKDT: the min/max KD-tree over cells the ray caster walks (src/SlimKDT.zig)."""
import ctypes as C
import os

import numpy as np

from . import capi


class SlimKDT:
    """nodes: uint32 heap-indexed array of packed SlimNode words (src/SlimKDT.zig:9-13);
    root_dim: uint8[3] = resolution - 1 (:129-133); build: PartitionBuildData (:215-219)."""

    def __init__(self, nodes, root_dim, build=None, data=None, wide=False):
        self.nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
        # wide: a volume beyond the reference's u8 coordinates; nodes are wide node words (12-bit partition, axis at
        # bit 28) and root_dim is uint16
        self.wide = bool(wide)
        self.root_dim = np.ascontiguousarray(root_dim, dtype=np.uint16 if self.wide else np.uint8)
        self.build = build
        self.data = data

    def saveToFile(self, file_name):
        """Tree cache file; the reference has only a stub for it (src/___SlimKDT_.zig:266-269)."""
        rd = (C.c_uint8 * 3)(*[int(x) for x in self.root_dim])
        capi.check(capi.lib().vr_save_kdt(os.fsencode(os.fspath(file_name)), self.nodes.ctypes.data, len(self.nodes), rd),
                   "vr_save_kdt")

    @staticmethod
    def loadFromFile(file_name):
        L = capi.lib()
        p = C.POINTER(C.c_uint32)()
        n = C.c_uint64()
        rd = (C.c_uint8 * 3)()
        capi.check(L.vr_load_kdt(os.fsencode(os.fspath(file_name)), C.byref(p), C.byref(n), rd), "vr_load_kdt")
        nodes = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        L.vr_free(p)
        return SlimKDT(nodes, np.array(list(rd), dtype=np.uint8))

    @staticmethod
    def left(i):
        return 2 * i + 1  # src/SlimKDT.zig:24-26

    @staticmethod
    def right(i):
        return 2 * i + 2  # src/SlimKDT.zig:28-30

    @staticmethod
    def parent(i):
        return (i - 1) // 2  # src/SlimKDT.zig:32-34

    @staticmethod
    def newHeursiticPartition(data, heuristic="AlwaysCenter", max_slots=0):
        """src/SlimKDT.zig:222-246, built on the host by libvr_hip.so.  Raises VrError(VR_ETOODEEP) where the
        reference prints "FAILD TO BUILD KDT"."""
        L = capi.lib()
        h = capi.HEURISTICS[heuristic] if isinstance(heuristic, str) else int(heuristic)
        res = (C.c_uint32 * 3)(*data.resulution)
        p = C.POINTER(C.c_uint32)()
        n = C.c_uint64()
        rd = (C.c_uint8 * 3)()
        st = capi.BuildStats()
        capi.check(L.vr_host_build_kdt(data.data.ctypes.data, res, h, max_slots, C.byref(p), C.byref(n), rd,
                                       C.byref(st)), "vr_host_build_kdt")
        nodes = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
        L.vr_free(p)
        return SlimKDT(nodes, np.array(list(rd), dtype=np.uint8), st, data)

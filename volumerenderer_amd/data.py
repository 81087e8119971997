"""Data: the 8-bit scalar volume of the reference (src/Data.zig)."""
import ctypes as C
import gzip
import os
import struct

import numpy as np

from . import capi


class Data:
    """resulution = (x, y, z) voxel counts; data = u8 array indexed [x, y, z] (x slowest, z fastest,
    src/Data.zig:90-92)."""

    def __init__(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        if data.ndim != 3:
            raise ValueError("volume must be 3-D")
        self.data = data
        self.resulution = tuple(int(x) for x in data.shape)

    @staticmethod
    def loadUCD(file_name):
        """src/Data.zig:9-44.  `.gz` files are gunzipped to a temporary image first."""
        file_name = os.fspath(file_name)
        if file_name.endswith(".gz"):
            raw = gzip.open(file_name, "rb").read()
            if len(raw) < 28:
                raise capi.VrError(capi.VR_EFORMAT, "Data.loadUCD")
            rx, ry, rz = struct.unpack(">3I", raw[:12])
            n = rx * ry * rz
            if n == 0 or len(raw) < 28 + n:
                raise capi.VrError(capi.VR_EFORMAT, "Data.loadUCD")
            return Data(np.frombuffer(raw, dtype=np.uint8, count=n, offset=28).reshape(rx, ry, rz))
        L = capi.lib()
        p = C.POINTER(C.c_uint8)()
        res = (C.c_uint32 * 3)()
        capi.check(L.vr_load_ucd(os.fsencode(file_name), C.byref(p), res), "vr_load_ucd")
        n = res[0] * res[1] * res[2]
        arr = np.ctypeslib.as_array(p, shape=(n,)).copy().reshape(res[0], res[1], res[2])
        L.vr_free(p)
        return Data(arr)

    def saveUCD(self, file_name):
        """Writes the volume in the layout loadUCD reads (the reference has a loader only)."""
        res = (C.c_uint32 * 3)(*self.resulution)
        vox = np.ascontiguousarray(self.data, dtype=np.uint8)
        capi.check(capi.lib().vr_save_ucd(os.fsencode(os.fspath(file_name)), vox.ctypes.data, res), "vr_save_ucd")

    def get(self, x, y, z):
        """src/Data.zig:90-92"""
        return int(self.data[x, y, z])

    def size(self):
        """src/Data.zig:98-100"""
        return int(self.data.size)

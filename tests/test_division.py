"""The MOVE step divides by a ray-direction component three times per skipped node.  The kernels replace the
IEEE divide by a per-ray reciprocal and two FMA residual corrections (vr_div in vr_trace_core.h); this checks
that replacement against the hardware divide, bit for bit, on 10^8 operand pairs of the range the path produces
(plane offsets 0..255 minus a camera coordinate, over direction components down to 1e-8)."""
import ctypes as C


def test_vr_div_equals_ieee_divide(emu_lib):
    emu_lib.vr_emu_div_check.restype = C.c_uint64
    emu_lib.vr_emu_div_check.argtypes = [C.c_uint64, C.c_uint64]
    assert emu_lib.vr_emu_div_check(100_000_000, 12345) == 0

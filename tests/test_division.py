"""The MOVE step divides by a ray-direction component three times per skipped node.  The kernels replace the
IEEE divide by a per-ray reciprocal and two FMA residual corrections (vr_div in vr_trace_core.h); this checks
that replacement against the hardware divide, bit for bit, on 10^8 operand pairs of the range the path produces
(plane offsets 0..255 minus a camera coordinate, over direction components down to 1e-8)."""
import ctypes as C

import pytest


def test_vr_div_equals_ieee_divide(emu_lib):
    emu_lib.vr_emu_div_check.restype = C.c_uint64
    emu_lib.vr_emu_div_check.argtypes = [C.c_uint64, C.c_uint64]
    assert emu_lib.vr_emu_div_check(100_000_000, 12345) == 0


def test_vr_div_equals_ieee_divide_on_wide_volumes(emu_lib):
    """Plane offsets 0..1023 (volumes of up to 1024 voxels per axis), camera coordinates up to 3000 away."""
    emu_lib.vr_emu_div_check_wide.restype = C.c_uint64
    emu_lib.vr_emu_div_check_wide.argtypes = [C.c_uint64, C.c_uint64]
    assert emu_lib.vr_emu_div_check_wide(100_000_000, 999) == 0


@pytest.mark.gpu
def test_gpu_fast_paths_equal_their_portable_definitions():
    """vr_selftest: the inline-assembly floor and carry paths and the reciprocal division of the MOVE, which the CPU
    emulation replaces by portable code, against those definitions ON the GPU (2^27 operand sets): no mismatch."""
    import ctypes as C
    import volumerenderer_amd as vr
    from volumerenderer_amd import capi
    with vr.Renderer(0) as r:
        bad = (C.c_uint64 * 3)(1, 1, 1)
        capi.check(capi.lib().vr_selftest(r.handle, 1 << 27, bad), "vr_selftest")
        assert list(bad) == [0, 0, 0], "mismatches (floor, carry, division): %s" % list(bad)

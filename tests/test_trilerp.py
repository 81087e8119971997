"""Trilinear refinement: trilerp (src/raytrace.zig:257-291) and getCornerDens (src/Data.zig:102-114).

CPU: the oracle's restatement against an independent float32 evaluation written from the reference's formula
(numpy rounds every float32 operation separately, like Zig's strict float mode), and the library's header
function (tests/emu build) against the oracle, bit for bit.
GPU: the kernel through the C ABI against the oracle, bit for bit, on random and degenerate arguments; the
refinement post-pass (corner voxels from device memory, trilerp at every hit point) against the oracle's.  Beyond
trilerp and getCornerDens the reference has nothing to compare the post-pass with: parity unpinned."""
import ctypes as C

import numpy as np
import pytest

import volumerenderer_amd as vr
from volumerenderer_amd import capi


def _records(rng, n):
    """n argument records: points inside and outside cells of mixed sizes, integer corner values as in the volume."""
    p = np.zeros((n, 17), np.float32)
    lo = rng.integers(0, 255, size=(n, 3)).astype(np.float32)
    size = np.where(rng.random((n, 3)) < 0.8, 1.0, rng.uniform(0.25, 7.0, size=(n, 3))).astype(np.float32)
    frac = rng.uniform(-0.25, 1.25, size=(n, 3)).astype(np.float32)
    p[:, 0:3] = lo + frac * size
    p[:, 3], p[:, 4] = lo[:, 0], lo[:, 0] + size[:, 0]
    p[:, 5], p[:, 6] = lo[:, 1], lo[:, 1] + size[:, 1]
    p[:, 7], p[:, 8] = lo[:, 2], lo[:, 2] + size[:, 2]
    p[:, 9:] = rng.integers(0, 256, size=(n, 8)).astype(np.float32)
    p[: n // 10, 9:] = rng.uniform(-1e3, 1e3, size=(n // 10, 8)).astype(np.float32)  # and arbitrary floats
    return p


def _numpy_trilerp(p):
    f = np.float32
    x, y, z, x0, x1, y0, y1, z0, z1 = [p[:, k] for k in range(9)]
    v = [p[:, 9 + k] for k in range(8)]
    xd, yd, zd = (x - x0) / (x1 - x0), (y - y0) / (y1 - y0), (z - z0) / (z1 - z0)
    one = f(1)
    c00 = v[0] * (one - xd) + v[1] * xd
    c10 = v[2] * (one - xd) + v[3] * xd
    c01 = v[4] * (one - xd) + v[5] * xd
    c11 = v[6] * (one - xd) + v[7] * xd
    c0 = c00 * (one - yd) + c10 * yd
    c1 = c01 * (one - yd) + c11 * yd
    return c0 * (one - zd) + c1 * zd


def test_oracle_trilerp_follows_the_reference_formula(oracle):
    p = _records(np.random.default_rng(1), 200000)
    got = oracle.trilerp_batch(p)
    assert (got.view(np.uint32) == _numpy_trilerp(p).view(np.uint32)).all()
    # corners reproduce their own values, the centre of a unit cell is the mean
    q = np.zeros((9, 17), np.float32)
    q[:, 3:9] = [10, 11, 20, 21, 30, 31]
    q[:, 9:] = [3, 7, 11, 19, 23, 101, 211, 255]
    for k in range(8):
        q[k, 0:3] = [10 + (k & 1), 20 + ((k >> 1) & 1), 30 + (k >> 2)]
    q[8, 0:3] = [10.5, 20.5, 30.5]
    out = oracle.trilerp_batch(q)
    assert list(out[:8]) == [3, 7, 11, 19, 23, 101, 211, 255] and out[8] == np.float32(630) / 8


def test_library_trilerp_and_corner_fetch_equal_the_oracle_on_the_cpu(oracle, emu_lib, c60_vox):
    p = _records(np.random.default_rng(2), 100000)
    out = np.zeros(len(p), np.float32)
    emu_lib.vr_emu_trilerp(C.c_void_p(p.ctypes.data), C.c_uint64(len(p)), C.c_void_p(out.ctypes.data))
    assert (out.view(np.uint32) == oracle.trilerp_batch(p).view(np.uint32)).all()
    # getCornerDens: v000, v100, v010, v110, v001, v101, v011, v111 with x slowest / z fastest
    rx, ry, rz = c60_vox.shape
    got = (C.c_float * 8)()
    for cx, cy, cz in [(0, 0, 0), (62, 62, 62), (17, 40, 5), (30, 31, 32)]:
        emu_lib.vr_emu_corner_dens(C.c_void_p(c60_vox.ctypes.data), C.c_uint32(ry), C.c_uint32(rz), C.c_uint32(cx), C.c_uint32(cy),
                                   C.c_uint32(cz), got)
        want = [c60_vox[cx + (k & 1), cy + ((k >> 1) & 1), cz + (k >> 2)] for k in range(8)]
        assert list(got) == [float(w) for w in want]


@pytest.mark.gpu
def test_gpu_trilerp_is_bit_exact(oracle):
    import torch
    rng = np.random.default_rng(3)
    p = _records(rng, 1 << 20)
    # degenerate cells (x1 == x0): the division gives inf / NaN on both sides
    p[:64, 4] = p[:64, 3]
    p[64:96, 0] = p[64:96, 3]
    p[64:96, 4] = p[64:96, 3]
    L = capi.lib()
    with vr.Renderer(0) as r:
        d_in = torch.from_numpy(p).cuda()
        d_out = torch.zeros(len(p), dtype=torch.float32, device="cuda")
        capi.check(L.vr_trilerp_device(r.handle, d_in.data_ptr(), len(p), d_out.data_ptr(), None), "vr_trilerp_device")
        capi.check(L.vr_sync(r.handle), "vr_sync")
        got = d_out.cpu().numpy()
    ref = oracle.trilerp_batch(p)
    nan = np.isnan(ref)
    assert (np.isnan(got) == nan).all() and nan.sum() >= 32  # NaN sign / payload is not part of the contract
    assert (got.view(np.uint32)[~nan] == ref.view(np.uint32)[~nan]).all()


@pytest.mark.gpu
def test_gpu_refinement_pass_equals_the_oracle(oracle, c60_vox, skull_vox):
    """vr_refine_device on the taps of a render: the interpolated density at every hit point, bit for bit the oracle's
    (parity unpinned beyond trilerp / getCornerDens); sanity: it lies inside the hit cell's corner range, which holds
    the iso value."""
    import torch
    L = capi.lib()
    for vox, W, H, isos in [(c60_vox, 200, 160, (30, 100, 222)), (skull_vox, 256, 192, (80,))]:
        pos, forw, up = oracle.main_camera(vox.shape)
        with vr.Renderer(0) as r:
            r.upload_volume(vr.Data(vox))
            r.build_kdt("AlwaysCenter")
            v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
            n = len(isos)
            out = [r.renderSweep(pos, forw, up, W, H, 2.0, iso, 1, taps=True) for iso in isos]
            cf = np.stack([o["cell_flag"][0] for o in out])
            tt = np.stack([o["t"][0] for o in out])
            d_cf, d_t = torch.from_numpy(cf).cuda(), torch.from_numpy(tt).cuda()
            d_den = torch.full((n, H, W), -1.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()  # the fills run on torch's stream, the library's kernels on the context's own
            capi.check(L.vr_refine_device(r.handle, C.byref(v), n, 0, H, d_cf.data_ptr(), d_t.data_ptr(), d_den.data_ptr(), None),
                       "vr_refine_device")
            capi.check(L.vr_sync(r.handle), "vr_sync")
            den = d_den.cpu().numpy()
            # a row range of one frame
            d_part = torch.full((1, 40, W), -1.0, dtype=torch.float32, device="cuda")
            a_cf, a_t = d_cf[0, 50:90].contiguous(), d_t[0, 50:90].contiguous()
            torch.cuda.synchronize()
            capi.check(L.vr_refine_device(r.handle, C.byref(v), 1, 50, 90, a_cf.data_ptr(), a_t.data_ptr(), d_part.data_ptr(), None), "vr_refine_device")
            capi.check(L.vr_sync(r.handle), "vr_sync")
            assert (d_part.cpu().numpy()[0].view(np.uint32) == den[0, 50:90].view(np.uint32)).all()
        for k, iso in enumerate(isos):
            ref = oracle.refine(vox, pos, forw, up, W, H, 2.0, cf[k], tt[k])
            assert (den[k].view(np.uint32) == ref.view(np.uint32)).all(), iso
            hit = (cf[k] >> 24) == capi.FLAG_HIT
            assert hit.sum() > 500 and (den[k][~hit] == 0).all()
            cx, cy, cz = cf[k] & 0xff, (cf[k] >> 8) & 0xff, (cf[k] >> 16) & 0xff
            corners = np.stack([vox[cx + (c & 1), cy + ((c >> 1) & 1), cz + (c >> 2)] for c in range(8)]).astype(np.float32)
            lo, hi = corners.min(0)[hit], corners.max(0)[hit]
            assert (lo <= iso).all() and (iso <= hi).all()
            assert (den[k][hit] >= lo - 0.01).all() and (den[k][hit] <= hi + 0.01).all()
    with vr.Renderer(0) as r:  # no volume on the context
        v = vr.raytrace._view((0, 0, 0), (0, 1, 0), (0, 0, 1), 8, 8, 2.0)
        d = torch.zeros(64, dtype=torch.float32, device="cuda")
        assert L.vr_refine_device(r.handle, C.byref(v), 1, 0, 8, d.data_ptr(), d.data_ptr(), d.data_ptr(), None) == capi.VR_ESTATE

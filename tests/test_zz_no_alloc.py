"""vr_render_device_noalloc: the device-resident render entry point whose pipeline scratch is a static device array
(no hipMalloc, no event creation).  Through the C ABI, against vr_render_device on the same inputs: frames, all
four taps and all seven per-iso counters bit for bit, on two calls in a row, the second with a smaller input (the
array keeps the first call's ray list and queue heads), for the sweep kernel, the one-traversal-per-iso kernel
and a wide tree; and the size limit is refused with VR_ELIMIT."""
import ctypes as C

import numpy as np
import pytest

import volumerenderer_amd as vr
from volumerenderer_amd import capi

pytestmark = pytest.mark.gpu


def _both(r, pos, forw, up, W, H, b, n, r0, r1, with_taps=True):
    """Renders rows [r0, r1) of n frames with both entry points; returns their outputs as numpy arrays.  Without taps a
    call of 8 or more iso values runs the kernel with one traversal per ray for the whole iso range, whose list of iso
    runs and counters lie in the scratch as well."""
    import torch
    L = capi.lib()
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    res = []
    for fn, name in ((L.vr_render_device, "vr_render_device"), (L.vr_render_device_noalloc, "vr_render_device_noalloc")):
        rows = r1 - r0
        d_rgb = torch.full((n, rows, W, 3), 9, dtype=torch.uint8, device="cuda")
        d_stats = torch.full((n, 7), -1, dtype=torch.int64, device="cuda")
        d_tap = [torch.full((n, rows, W), 5, dtype=torch.int32, device="cuda") for _ in range(4 if with_taps else 0)]
        taps = capi.Taps(*[t.data_ptr() for t in d_tap]) if with_taps else None
        torch.cuda.synchronize()
        capi.check(fn(r.handle, C.byref(v), b, n, r0, r1, d_rgb.data_ptr(), d_stats.data_ptr(), C.byref(taps) if with_taps else None, None), name)
        capi.check(L.vr_sync(r.handle), "vr_sync")
        res.append([d_rgb.cpu().numpy(), d_stats.cpu().numpy()] + [t.cpu().numpy() for t in d_tap])
    return res


def _assert_equal(res, what):
    a, b = res
    for i, name in enumerate(("rgb", "stats", "cell_flag", "t", "checked", "node")[:len(a)]):
        assert (a[i] == b[i]).all(), "%s: %s differs" % (what, name)
    assert a[1][:, 1].sum() > 0 and (a[1][:, 6] == 0).all(), what  # something was hit, no errors


def test_noalloc_entry_point_without_taps_runs_the_union_kernel(oracle, c60_vox, c60_tree, skull_vox, skull_tree):
    """The normal serving call -- no taps, 8 or more iso values -- goes to the kernel with one traversal per ray, its second
    kernel and its list in the static array: frames and all 7 counters equal vr_render_device's, a large call and then a
    small one, for the default selection and with the sweep / union kernels named; one frame set also against the oracle."""
    for vox, tree, cases in ((skull_vox, skull_tree, [(512, 512, 40, 64, 0, 512), (200, 160, 70, 12, 31, 140)]),
                             (c60_vox, c60_tree, [(256, 256, 0, 256, 0, 256), (64, 48, 96, 9, 5, 40)])):
        nodes, rd, _ = tree
        pos, forw, up = oracle.main_camera(vox.shape)
        with vr.Renderer(0) as r:
            r.upload_kdt(vr.SlimKDT(nodes, rd))
            for variant in (capi.KERNEL_DEFAULT, capi.KERNEL_UNION, capi.KERNEL_SWEEP):
                r.set_kernel(variant)
                for (W, H, b, n, r0, r1) in cases:
                    res = _both(r, pos, forw, up, W, H, b, n, r0, r1, with_taps=False)
                    _assert_equal(res, "%s %dx%d x %d iso, kernel %d, no taps" % (vox.shape, W, H, n, variant))
            W, H, b, n, r0, r1 = cases[1]
            for j in (0, n - 1):
                ref = oracle.render(nodes, rd, pos, forw, up, W, H, 2.0, b + j, rows=(r0, r1))
                assert (res[1][0][j] == ref["rgb"][r0:r1]).all() and res[1][1][j, 0] == ref["stats"].nodes_checked
    rng = np.random.default_rng(6)
    vox = (rng.random((300, 20, 24)) * 255).astype(np.uint8)
    with vr.Renderer(0) as r:  # a wide tree, 8 or more iso values
        r.upload_volume(vr.Data(vox))
        r.build_kdt("AlwaysCenter")
        pos, forw, up = oracle.main_camera(vox.shape)
        _assert_equal(_both(r, pos, forw, up, 160, 120, 90, 40, 0, 120, with_taps=False), "wide 160x120 x 40 iso, no taps")
        _assert_equal(_both(r, pos, forw, up, 64, 48, 100, 8, 8, 40, with_taps=False), "wide 64x48 rows 8..40 x 8 iso, no taps")


def test_noalloc_entry_point_equals_vr_render_device(oracle, c60_vox, c60_tree, skull_vox, skull_tree):
    nodes, rd, _ = skull_tree
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    with vr.Renderer(0) as r:
        r.upload_kdt(vr.SlimKDT(nodes, rd))
        # sweep kernel: a large call, then a smaller one (fewer pixels, fewer iso values, a row range) right behind it
        _assert_equal(_both(r, pos, forw, up, 512, 512, 40, 64, 0, 512), "skull 512^2 x 64 iso")
        _assert_equal(_both(r, pos, forw, up, 200, 160, 70, 12, 31, 140), "skull 200x160 rows 31..140 x 12 iso")
        # one traversal per iso value (fewer than 8 iso values), again large then small
        _assert_equal(_both(r, pos, forw, up, 1024, 1024, 80, 1, 0, 1024), "skull 1024^2 iso 80")
        _assert_equal(_both(r, pos, forw, up, 96, 64, 80, 3, 0, 64), "skull 96x64 x 3 iso")
        # and against the oracle, so that both being wrong the same way cannot pass
        got = _both(r, pos, forw, up, 96, 64, 80, 1, 0, 64)[1]
        ref = oracle.render(nodes, rd, pos, forw, up, 96, 64, 2.0, 80)
        assert (got[0][0] == ref["rgb"]).all() and (got[4][0] == ref["checked"]).all()
        assert (got[2][0].view(np.uint32) == ref["cell_flag"]).all() and got[1][0, 0] == ref["stats"].nodes_checked
        # the limit: more than 2^20 pixels per call is refused, nothing is launched
        import torch
        L = capi.lib()
        v = vr.raytrace._view(pos, forw, up, 1024, 1025, 2.0)
        d = torch.zeros(8, dtype=torch.uint8, device="cuda")
        assert L.vr_render_device_noalloc(r.handle, C.byref(v), 80, 1, 0, 1025, d.data_ptr(), None, None, None) == capi.VR_ELIMIT
    # a second context on the same GPU shares the array: a different tree, right after the calls above
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    with vr.Renderer(0) as r:
        r.upload_kdt(vr.SlimKDT(nodes, rd))
        _assert_equal(_both(r, pos, forw, up, 256, 256, 0, 256, 0, 256), "c60 256^2 x 256 iso")
        _assert_equal(_both(r, pos, forw, up, 37, 53, 100, 1, 0, 53), "c60 37x53 iso 100")


def test_noalloc_entry_point_on_a_wide_tree(oracle):
    """A volume beyond 256 voxels per axis (wide node words, the wide persistent kernel) through the same entry point."""
    rng = np.random.default_rng(5)
    vox = (rng.random((300, 20, 24)) * 255).astype(np.uint8)
    with vr.Renderer(0) as r:
        r.upload_volume(vr.Data(vox))
        r.build_kdt("AlwaysCenter")
        pos, forw, up = oracle.main_camera(vox.shape)
        _assert_equal(_both(r, pos, forw, up, 160, 120, 100, 2, 0, 120), "wide 160x120")
        _assert_equal(_both(r, pos, forw, up, 64, 48, 100, 1, 8, 40), "wide 64x48 rows 8..40")

"""The PNG / PPM writers of the library (the output side of the reference driver, src/main.zig:116,137)."""
import numpy as np
import pytest

from volumerenderer_amd import capi

PIL = pytest.importorskip("PIL.Image")


def test_png_and_ppm_round_trip(tmp_path, golden_gray):
    L = capi.lib()
    for iso, (w, h) in [(100, (512, 512)), (0, (512, 512))]:
        rgb = np.ascontiguousarray(np.repeat(golden_gray[iso][..., None], 3, axis=2))
        for name, fn in (("a.png", L.vr_write_png_rgb24), ("a.ppm", L.vr_write_ppm_rgb24)):
            p = tmp_path / name
            assert fn(str(p).encode(), w, h, rgb.ctypes.data) == 0
            assert (np.asarray(PIL.open(p).convert("RGB")) == rgb).all()
    # ragged sizes, more than one stored deflate block (> 65535 bytes) and a single-pixel image
    rng = np.random.default_rng(1)
    for w, h in [(1, 1), (37, 53), (300, 91)]:
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        p = tmp_path / ("r%dx%d.png" % (w, h))
        assert L.vr_write_png_rgb24(str(p).encode(), w, h, rgb.ctypes.data) == 0
        assert (np.asarray(PIL.open(p).convert("RGB")) == rgb).all()
    assert L.vr_write_png_rgb24(None, 1, 1, None) == capi.VR_EARG
    assert L.vr_write_png_rgb24(str(tmp_path / "no" / "dir.png").encode(), 1, 1, rgb.ctypes.data) == capi.VR_EIO

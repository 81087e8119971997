"""The PNG / PPM writers of the library (the output side of the reference driver, src/main.zig:116,137)."""
import numpy as np
import pytest

import volumerenderer_amd as vr
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


def test_vol_writer_round_trip(tmp_path, c60_vox):
    """vr_save_ucd writes what Data.loadUCD (src/Data.zig:9-44) reads; the file of the reference's own
    C60 volume comes back byte for byte (its three ignored "true size" floats aside)."""
    import os
    rng = np.random.default_rng(5)
    for vox in (c60_vox, rng.integers(0, 256, (3, 7, 5), dtype=np.uint8)):
        path = tmp_path / "v.vol"
        vr.Data(vox).saveUCD(path)
        back = vr.Data.loadUCD(path)
        assert back.resulution == vox.shape and (back.data == vox).all()
    ref = open(os.path.join(os.path.dirname(__file__), "data", "C60.vol"), "rb").read()
    vr.Data(c60_vox).saveUCD(tmp_path / "c60.vol")
    mine = open(tmp_path / "c60.vol", "rb").read()
    assert len(mine) == len(ref) and mine[:16] == ref[:16] and mine[28:] == ref[28:]


def test_tree_cache_file_round_trip(tmp_path, c60_vox, c60_tree):
    nodes, rd, _ = c60_tree
    k = vr.SlimKDT(nodes, rd)
    path = tmp_path / "c60.kdt"
    k.saveToFile(path)
    back = vr.SlimKDT.loadFromFile(path)
    assert len(back.nodes) == len(nodes) and (back.nodes == nodes).all() and (back.root_dim == rd).all()
    import os
    padded = vr.SlimKDT(np.concatenate([nodes, np.zeros(len(nodes), np.uint32)]), rd)
    padded.saveToFile(tmp_path / "padded.kdt")  # trailing unoccupied slots are not stored, but come back
    assert os.path.getsize(tmp_path / "padded.kdt") == os.path.getsize(path)
    back2 = vr.SlimKDT.loadFromFile(tmp_path / "padded.kdt")
    assert len(back2.nodes) == 2 * len(nodes) and (back2.nodes[: len(nodes)] == nodes).all() and not back2.nodes[len(nodes):].any()
    raw = bytearray(open(path, "rb").read())
    raw[100] ^= 1
    open(path, "wb").write(raw)
    with pytest.raises(capi.VrError) as e:
        vr.SlimKDT.loadFromFile(path)
    assert e.value.code == capi.VR_EFORMAT
    open(path, "wb").write(b"not a tree")
    with pytest.raises(capi.VrError):
        vr.SlimKDT.loadFromFile(path)

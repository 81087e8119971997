"""GPU parity for volumes beyond the reference's u8 coordinates (up to 1024 voxels per axis).

No reference artefact exists for such volumes (SPACESIZE = u8, src/typedef.zig:1): parity is UNPINNED beyond
what these tests tie together -- the wide kernel equals the pinned u8 path (oracle and the reference's own
renders) on volumes both can represent, and equals the u16 build of the same oracle file on wide volumes.
At the full 1024^3 size the oracle cannot run; size-independent properties stand in."""
import ctypes as C

import numpy as np
import pytest

import volumerenderer_amd as vr
from volumerenderer_amd import capi, datasets
from helpers import assert_same_render, random_volume
from test_wide_cpu import WIDE_SHAPES, narrow_to_wide_cell_flag, narrow_to_wide_words

pytestmark = pytest.mark.gpu

# wide trees: the pipeline (compacted ray list + persistent waves with lane refill) and the per-pixel kernel
WIDE_KERNELS = [capi.KERNEL_DEFAULT, capi.KERNEL_RESUME]


def _gpu(renderer, pos, forw, up, w, h, vp, iso):
    img = vr.Image(w, h)
    res = renderer.renderImage(pos, forw, up, img, vp, iso, taps=True)
    res["rgb"] = img.pixels
    return res


def test_wide_kernel_on_c60_matches_reference_renders_and_u8_oracle(oracle, c60_vox, c60_tree, golden_gray):
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    with vr.Renderer(0) as r:
        r.upload_kdt(vr.SlimKDT(narrow_to_wide_words(nodes), rd, wide=True))
        out = r.renderSweep(pos, forw, up, 512, 512, 2.0, 0, 256)
        for iso in range(256):
            assert (out["rgb"][iso][:, :, 0] == golden_gray[iso]).all(), iso
        r.set_kernel(capi.KERNEL_RESUME)
        per_pixel = r.renderSweep(pos, forw, up, 512, 512, 2.0, 0, 256)
        assert (per_pixel["rgb"] == out["rgb"]).all() and per_pixel["stats"] == out["stats"]
        for iso in (30, 100, 200):
            ref = oracle.render(nodes, rd, pos, forw, up, 200, 160, 2.0, iso)
            ref["cell_flag"] = narrow_to_wide_cell_flag(ref["cell_flag"])
            for kern in WIDE_KERNELS:
                r.set_kernel(kern)
                got = _gpu(r, pos, forw, up, 200, 160, 2.0, iso)
                assert_same_render(got, ref, "wide kernel %d on C60 iso %d" % (kern, iso))
                assert got["stats"]["nodes_checked"] == ref["stats"].nodes_checked and got["stats"]["errors"] == 0


@pytest.mark.parametrize("shape,kind", WIDE_SHAPES + [((1024, 5, 3), "smooth"), ((400, 300, 3), "noise")])
def test_wide_volumes_gpu_build_and_render_match_wide_oracle(oracle, shape, kind):
    rng = np.random.default_rng(sum(shape))
    vox = random_volume(rng, shape, kind)
    nodes, rd, st = oracle.build_kdt(vox, wide=True)
    with vr.Renderer(0) as r:
        r.upload_volume(vr.Data(vox))
        bs = r.build_kdt("AlwaysCenter")
        assert (bs.number_of_nodes, bs.layer_count, bs.non_center_partition_count) == (st.number_of_nodes, st.layer_count, 0)
        got_nodes, got_rd, n_dev = r.download_kdt_wide()
        assert (got_rd == rd).all() and n_dev == len(got_nodes)
        m = min(len(nodes), len(got_nodes))
        assert (got_nodes[:m] == nodes[:m]).all() and not got_nodes[m:].any() and not nodes[m:].any()
        window, _, _ = r.download_kdt_wide(3, 5)
        assert (window == nodes[3:8]).all()
        cams = [(-30.0, -300.0, -30.0), (float(shape[0]) + 40.0, 20.0, 10.0), (8.0, 8.0, -320.0)]
        for pos in cams:
            pos, forw, up = oracle.main_camera(shape, pos)
            for vp, iso in ((0.2, 0), (3.0, 90), (0.7, 200)):
                ref = oracle.render(nodes, rd, pos, forw, up, 70, 50, vp, iso, wide=True)
                for kern in WIDE_KERNELS:
                    r.set_kernel(kern)
                    got = _gpu(r, pos, forw, up, 70, 50, vp, iso)
                    assert_same_render(got, ref, "%s %s iso %d kernel %d" % (shape, pos, iso, kern))
                    s = got["stats"]
                    assert (s["nodes_checked"], s["hit_count"], s["miss_count"], s["oob_count"], s["moves"], s["errors"]) == (
                        ref["stats"].nodes_checked, ref["stats"].hit_count, ref["stats"].miss_count, ref["stats"].oob_count,
                        ref["stats"].moves, 0)
                r.set_kernel(capi.KERNEL_DEFAULT)
        # the uploaded form of the same tree renders the same
        pos, forw, up = oracle.main_camera(shape)
        a = _gpu(r, pos, forw, up, 64, 64, 1.0, 120)
        r.upload_kdt(vr.SlimKDT(nodes, rd, wide=True))
        b = _gpu(r, pos, forw, up, 64, 64, 1.0, 120)
        assert_same_render(b, a, "uploaded vs built")


def test_wide_sweep_and_row_shards(oracle):
    import torch
    from volumerenderer_amd import sharding
    vox = random_volume(np.random.default_rng(3), (520, 40, 36), "smooth")
    nodes, rd, _ = oracle.build_kdt(vox, wide=True)
    pos, forw, up = oracle.main_camera(vox.shape)
    W, H = 150, 90
    L = capi.lib()
    with vr.Renderer(0) as r:
        r.upload_kdt(vr.SlimKDT(nodes, rd, wide=True))
        out = r.renderSweep(pos, forw, up, W, H, 2.5, 96, 6, taps=True)
        for j in range(6):
            ref = oracle.render(nodes, rd, pos, forw, up, W, H, 2.5, 96 + j, wide=True)
            got = {k: out[k][j] for k in ("rgb", "cell_flag", "t", "checked", "node")}
            assert_same_render(got, ref, "wide sweep iso %d" % (96 + j))
            assert out["stats"][j]["nodes_checked"] == ref["stats"].nodes_checked
        v = vr.raytrace._view(pos, forw, up, W, H, 2.5)
        total = 0
        for rank in range(3):
            r0, r1 = sharding.row_shard(rank, 3, H)
            d_rgb = torch.zeros((6, r1 - r0, W, 3), dtype=torch.uint8, device="cuda")
            d_stats = torch.zeros((6, 7), dtype=torch.int64, device="cuda")
            capi.check(L.vr_render_device(r.handle, C.byref(v), 96, 6, r0, r1, d_rgb.data_ptr(), d_stats.data_ptr(), None,
                                          None), "vr_render_device")
            capi.check(L.vr_sync(r.handle), "vr_sync")
            assert (d_rgb.cpu().numpy() == out["rgb"][:, r0:r1]).all(), rank
            total += int(d_stats.cpu().numpy()[:, 0].sum())
        assert total == sum(s["nodes_checked"] for s in out["stats"])


def test_wide_sweep_kernel_every_iso_value_matches_wide_oracle(oracle):
    """The sweep kernel with the wide lane code (one traversal for every iso value that cannot change a range test on
    its way): all 256 iso values of volumes with coordinates beyond 255 against one oracle render each -- pixels,
    cell, t bits, hit leaf, reference-defined check count, all 7 counters -- and, at a size where waves run full, the
    frames and counters of the one-traversal-per-iso kernel."""
    import torch
    rng = np.random.default_rng(8)
    for shape, kind, W, H in [((300, 24, 20), "noise", 72, 56), ((40, 520, 36), "smooth", 96, 64), ((33, 20, 700), "sparse", 64, 48)]:
        vox = random_volume(rng, shape, kind)
        nodes, rd, _ = oracle.build_kdt(vox, wide=True)
        pos, forw, up = oracle.main_camera(vox.shape)
        with vr.Renderer(0) as r:
            r.upload_kdt(vr.SlimKDT(nodes, rd, wide=True))
            r.set_kernel(capi.KERNEL_SWEEP)
            out = r.renderSweep(pos, forw, up, W, H, 2.5, 0, 256, taps=True)
        hits = 0
        for j in range(0, 256, 2):
            ref = oracle.render(nodes, rd, pos, forw, up, W, H, 2.5, j, wide=True)
            got = {k: out[k][j] for k in ("rgb", "cell_flag", "t", "checked", "node")}
            assert_same_render(got, ref, "wide sweep kernel %s iso %d" % (shape, j))
            st, o = out["stats"][j], ref["stats"]
            assert (st["nodes_checked"], st["hit_count"], st["miss_count"], st["oob_count"], st["escape_help_events"], st["moves"],
                    st["errors"]) == (o.nodes_checked, o.hit_count, o.miss_count, o.oob_count, o.escape_help_events, o.moves, 0)
            hits += o.hit_count
        assert hits > 1000
    vox = datasets.gyroid_noise(320)
    pos, forw, up = oracle.main_camera(vox.shape)
    L = capi.lib()
    W = H = 768
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    with vr.Renderer(0) as r:
        r.upload_volume(vr.Data(vox))
        r.build_kdt("AlwaysCenter")
        res = []
        for variant in (capi.KERNEL_PIPELINE, capi.KERNEL_SWEEP):
            r.set_kernel(variant)
            d_rgb = torch.empty((64, H, W, 3), dtype=torch.uint8, device="cuda")
            d_stats = torch.zeros((64, 7), dtype=torch.int64, device="cuda")
            capi.check(L.vr_render_device(r.handle, C.byref(v), 100, 64, 0, H, d_rgb.data_ptr(), d_stats.data_ptr(), None, None),
                       "vr_render_device")
            capi.check(L.vr_sync(r.handle), "vr_sync")
            res.append((d_rgb, d_stats.cpu().numpy()))
    assert (res[0][1] == res[1][1]).all() and int(res[0][1][:, 1].sum()) > 10 ** 6 and int(res[0][1][:, 6].sum()) == 0
    assert bool(torch.equal(res[0][0], res[1][0]))


def test_wide_union_kernel_every_iso_value_matches_wide_oracle(oracle):
    """ONE traversal per ray for the whole iso range on wide trees (vr_k_union with the wide lane code, the default for
    sweeps without taps): frames and all 7 counters of every iso value against one oracle render each, ranges that do not
    start at 0, and at a size where waves run full the frames and counters of the one-traversal-per-iso kernel."""
    import torch
    rng = np.random.default_rng(8)
    for shape, kind, W, H, b, n in [((300, 24, 20), "noise", 72, 56, 0, 256), ((40, 520, 36), "smooth", 96, 64, 0, 256),
                                    ((33, 20, 700), "sparse", 64, 48, 0, 256), ((300, 24, 20), "noise", 50, 40, 77, 9)]:
        vox = random_volume(rng, shape, kind)
        nodes, rd, _ = oracle.build_kdt(vox, wide=True)
        pos, forw, up = oracle.main_camera(vox.shape)
        with vr.Renderer(0) as r:
            r.upload_kdt(vr.SlimKDT(nodes, rd, wide=True))
            r.set_kernel(capi.KERNEL_UNION)
            out = r.renderSweep(pos, forw, up, W, H, 2.5, b, n)
        hits = 0
        for j in range(0, n, 2 if n > 50 else 1):
            ref = oracle.render(nodes, rd, pos, forw, up, W, H, 2.5, b + j, wide=True)
            assert (out["rgb"][j] == ref["rgb"]).all(), "wide union kernel %s iso %d" % (shape, b + j)
            st, o = out["stats"][j], ref["stats"]
            assert (st["nodes_checked"], st["hit_count"], st["miss_count"], st["oob_count"], st["escape_help_events"], st["moves"],
                    st["errors"]) == (o.nodes_checked, o.hit_count, o.miss_count, o.oob_count, o.escape_help_events, o.moves, 0), (shape, b + j)
            hits += o.hit_count
        assert hits > 500
    vox = datasets.gyroid_noise(320)
    pos, forw, up = oracle.main_camera(vox.shape)
    L = capi.lib()
    W = H = 768
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    with vr.Renderer(0) as r:
        r.upload_volume(vr.Data(vox))
        r.build_kdt("AlwaysCenter")
        res = []
        for variant in (capi.KERNEL_PIPELINE, capi.KERNEL_UNION, capi.KERNEL_DEFAULT):
            r.set_kernel(variant)
            d_rgb = torch.empty((64, H, W, 3), dtype=torch.uint8, device="cuda")
            d_stats = torch.zeros((64, 7), dtype=torch.int64, device="cuda")
            capi.check(L.vr_render_device(r.handle, C.byref(v), 100, 64, 0, H, d_rgb.data_ptr(), d_stats.data_ptr(), None, None),
                       "vr_render_device")
            capi.check(L.vr_sync(r.handle), "vr_sync")
            res.append((d_rgb, d_stats.cpu().numpy()))
        si = capi.SweepInfo()
        capi.check(L.vr_sweep_info_read(r.handle, C.byref(si)), "vr_sweep_info_read")
        assert 0 < si.walks <= si.items  # the default is the union kernel: one traversal per entering ray and block of iso values
    for k in (1, 2):
        assert (res[0][1] == res[k][1]).all(), np.argwhere(res[0][1] != res[k][1])[:8]
        assert bool(torch.equal(res[0][0], res[k][0]))
    assert int(res[0][1][:, 1].sum()) > 10 ** 6 and int(res[0][1][:, 6].sum()) == 0


def test_wide_malformed_tree_is_refused_at_upload():
    rng = np.random.default_rng(5)
    nodes = rng.integers(0, 2 ** 32, 4096, dtype=np.uint64).astype(np.uint32)
    with vr.Renderer(0) as r:
        with pytest.raises(capi.VrError) as e:
            r.upload_kdt(vr.SlimKDT(nodes, np.array([300, 200, 100]), wide=True))
        assert e.value.code == capi.VR_ETREE


def test_wide_argument_checks():
    L = capi.lib()
    with vr.Renderer(0) as r:
        vox = np.zeros((2, 2, 1025), np.uint8)
        assert L.vr_upload_volume(r.handle, vox.ctypes.data, 2, 2, 1025) == capi.VR_EARG
        vox = np.zeros((300, 4, 4), np.uint8)
        r.upload_volume(vr.Data(vox))
        assert L.vr_build_kdt(r.handle, capi.HEURISTICS["MinimizeSpan"], None) == capi.VR_EARG  # centre split only
        r.build_kdt("AlwaysCenter")
        n = C.c_uint64()
        assert L.vr_download_kdt(r.handle, None, 0, C.byref(n), None) == capi.VR_ESTATE  # wide tree: other call
        rd = (C.c_uint16 * 3)(1024, 4, 4)
        one = np.ones(8, np.uint32)
        assert L.vr_upload_kdt_wide(r.handle, one.ctypes.data, 2 ** 31 + 1, rd) == capi.VR_EARG
        assert L.vr_upload_kdt_wide(r.handle, one.ctypes.data, 8, rd) == capi.VR_EARG


def test_gyroid_1024_cubed_full_size_properties():
    """The 1024^3 stress configuration at full size: 2^31-slot tree built on the GPU (8 GiB), 4096 x 4096
    frame.  The CPU oracle cannot run at this size; checked instead: node count, the root's range = range of
    the whole volume, every hit cell's 8 voxel corners bracket the iso value and equal the hit leaf's stored
    range, 4 row shards = the whole frame with the same counters, a two-iso call = two one-iso calls."""
    import torch
    from volumerenderer_amd import sharding
    n = 1024
    vox = datasets.gyroid_noise(n)
    L = capi.lib()
    with vr.Renderer(0) as r:
        r.upload_volume(vr.Data(vox))
        bs = r.build_kdt("AlwaysCenter")
        assert bs.number_of_nodes == 2 * (n - 1) ** 3 - 1 and bs.layer_count == 30
        root, rd, n_dev = r.download_kdt_wide(0, 1)
        assert n_dev == 2 ** 31 and (rd == n - 1).all()
        assert (int(root[0]) & 0xff, (int(root[0]) >> 8) & 0xff) == (int(vox.min()), int(vox.max()))
        assert (int(root[0]) >> 16) & 0xfff == (n - 1) // 2 and (int(root[0]) >> 28) & 3 == 2
        pos, forw, up = vr.main_camera(vox.shape)
        W = H = 4096
        iso = 140
        whole = r.renderSweep(pos, forw, up, W, H, 2.0, iso, 1, taps=True)
        ws = whole["stats"][0]
        assert ws["errors"] == 0 and ws["hit_count"] > 100000
        assert ws["hit_count"] + ws["miss_count"] + ws["oob_count"] == W * H
        cf = whole["cell_flag"][0]
        hit = (cf >> 30) == capi.FLAG_HIT
        assert int(hit.sum()) == ws["hit_count"]
        x, y, z = (cf[hit] & 1023).astype(np.int64), ((cf[hit] >> 10) & 1023).astype(np.int64), ((cf[hit] >> 20) & 1023).astype(np.int64)
        corners = np.stack([vox[x + (k & 1), y + ((k >> 1) & 1), z + (k >> 2)] for k in range(8)])
        assert (corners.min(0) <= iso).all() and (iso <= corners.max(0)).all()
        leaf = whole["node"][0][hit]
        pick = np.random.default_rng(0).choice(len(leaf), 64, replace=False)
        for j in pick:
            word, _, _ = r.download_kdt_wide(int(leaf[j]), 1)
            assert (int(word[0]) & 0xff, (int(word[0]) >> 8) & 0xff) == (int(corners[:, j].min()), int(corners[:, j].max()))
        # rays that hit carry t inside the shading window or beyond it; pixels follow src/raytrace.zig:56-61
        t = whole["t"][0][hit]
        g = 255 - np.clip(np.trunc(np.minimum(200.0, t - np.float32(300.0))), 0, 255).astype(np.uint8)
        assert (whole["rgb"][0][hit][:, 0] == g).all() and not whole["rgb"][0][~hit].any()
        v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
        stats = np.zeros(7, np.int64)
        for rank in range(4):
            r0, r1 = sharding.row_shard(rank, 4, H)
            d_rgb = torch.zeros((1, r1 - r0, W, 3), dtype=torch.uint8, device="cuda")
            d_stats = torch.zeros((1, 7), dtype=torch.int64, device="cuda")
            capi.check(L.vr_render_device(r.handle, C.byref(v), iso, 1, r0, r1, d_rgb.data_ptr(), d_stats.data_ptr(), None,
                                          None), "vr_render_device")
            capi.check(L.vr_sync(r.handle), "vr_sync")
            assert (d_rgb.cpu().numpy()[0] == whole["rgb"][0, r0:r1]).all(), rank
            stats += d_stats.cpu().numpy()[0]
        assert [int(s) for s in stats] == [ws[k] for k in ("nodes_checked", "hit_count", "miss_count", "oob_count",
                                                            "escape_help_events", "moves", "errors")]
        r.set_kernel(capi.KERNEL_RESUME)  # the per-pixel kernel renders the same frame, counter for counter
        other = r.renderSweep(pos, forw, up, W, H, 2.0, iso, 1)
        assert (other["rgb"] == whole["rgb"]).all() and other["stats"][0] == ws
        r.set_kernel(capi.KERNEL_DEFAULT)
        two = r.renderSweep(pos, forw, up, 1024, 1024, 2.0, iso, 2)
        for j in range(2):
            one = r.renderSweep(pos, forw, up, 1024, 1024, 2.0, iso + j, 1)
            assert (one["rgb"][0] == two["rgb"][j]).all() and one["stats"][0] == two["stats"][j]

"""GPU parity: libvr_hip.so through its C ABI against the CPU oracle and the reference's committed renders.

Bar: bit-exact on every per-ray output -- RGB bytes, hit cell index, flag, t bit pattern, hit leaf index and
the reference-defined checked_nodes count -- for every traversal kernel variant."""
import ctypes as C

import numpy as np
import pytest

import helpers

import volumerenderer_amd as vr
from volumerenderer_amd import capi
from helpers import assert_same_render, random_volume

pytestmark = pytest.mark.gpu

VARIANTS = [capi.KERNEL_RESTART_GLOBAL, capi.KERNEL_RESTART_LDS, capi.KERNEL_RESUME, capi.KERNEL_PIPELINE,
            capi.KERNEL_SWEEP]


@pytest.fixture(scope="module")
def c60_renderer(c60_tree):
    nodes, rd, _ = c60_tree
    r = vr.Renderer(0)
    r.upload_kdt(vr.SlimKDT(nodes, rd))
    yield r
    r.close()


@pytest.fixture(scope="module")
def skull_renderer(skull_tree):
    nodes, rd, _ = skull_tree
    r = vr.Renderer(0)
    r.upload_kdt(vr.SlimKDT(nodes, rd))
    yield r
    r.close()


@pytest.fixture(scope="module")
def standin_tree():
    vox = vr.datasets.bonsai_standin()
    return vox, vr.SlimKDT.newHeursiticPartition(vr.Data(vox), "AlwaysCenter")


@pytest.fixture(scope="module")
def standin_renderer(standin_tree):
    """The 256^3 stand-in with its tree built on the GPU (word for word the host-built one:
    test_gpu_tree_build_matches_host_builder)."""
    r = vr.Renderer(0)
    r.upload_volume(vr.Data(standin_tree[0]))
    r.build_kdt("AlwaysCenter")
    yield r
    r.close()


def _gpu(renderer, pos, forw, up, w, h, vp, iso):
    img = vr.Image(w, h)
    res = renderer.renderImage(pos, forw, up, img, vp, iso, taps=True)
    res["rgb"] = img.pixels
    return res


def _check_stats(res, ref):
    st, o = res["stats"], ref["stats"]
    assert st["nodes_checked"] == o.nodes_checked
    assert st["hit_count"] == o.hit_count and st["miss_count"] == o.miss_count and st["oob_count"] == o.oob_count
    assert st["escape_help_events"] == o.escape_help_events and st["moves"] == o.moves
    assert st["errors"] == 0


@pytest.mark.parametrize("variant", VARIANTS)
def test_c60_512_matches_oracle(oracle, c60_vox, c60_tree, c60_renderer, variant):
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    c60_renderer.set_kernel(variant)
    for iso in (0, 1, 100, 255):
        ref = oracle.render(nodes, rd, pos, forw, up, 512, 512, 2.0, iso)
        got = _gpu(c60_renderer, pos, forw, up, 512, 512, 2.0, iso)
        assert_same_render(got, ref, "C60 512 iso %d variant %d" % (iso, variant))
        _check_stats(got, ref)
    c60_renderer.set_kernel(capi.KERNEL_DEFAULT)


def test_c60_sweep_matches_all_256_reference_renders(oracle, c60_vox, c60_renderer, golden_gray):
    """The reference's own artefacts: renders/out_iso{0..255}.png, as one vr_render_sweep call."""
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    out = c60_renderer.renderSweep(pos, forw, up, 512, 512, 2.0, 0, 256)
    rgb = out["rgb"]
    assert (rgb[..., 0] == golden_gray).all(), "%d pixels differ" % int((rgb[..., 0] != golden_gray).sum())
    assert (rgb[..., 0] == rgb[..., 1]).all() and (rgb[..., 1] == rgb[..., 2]).all()
    assert all(s["oob_count"] == 216422 and s["errors"] == 0 for s in out["stats"])


def test_c60_1024_matches_oracle(oracle, c60_vox, c60_tree, c60_renderer):
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    ref = oracle.render(nodes, rd, pos, forw, up, 1024, 1024, 2.0, 100)
    got = _gpu(c60_renderer, pos, forw, up, 1024, 1024, 2.0, 100)
    assert_same_render(got, ref, "C60 1024 iso 100")
    _check_stats(got, ref)
    assert ref["stats"].nodes_checked == 13759010  # the value the survey measured for this configuration


@pytest.mark.parametrize("variant", VARIANTS)
def test_skull_matches_oracle(oracle, skull_vox, skull_tree, skull_renderer, variant):
    nodes, rd, _ = skull_tree
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    skull_renderer.set_kernel(variant)
    for iso in (0, 80, 255):
        ref = oracle.render(nodes, rd, pos, forw, up, 384, 384, 2.0, iso)
        got = _gpu(skull_renderer, pos, forw, up, 384, 384, 2.0, iso)
        assert_same_render(got, ref, "Skull 384 iso %d variant %d" % (iso, variant))
        _check_stats(got, ref)
    skull_renderer.set_kernel(capi.KERNEL_DEFAULT)


def test_skull_1024_iso80_matches_oracle(oracle, skull_vox, skull_tree, skull_renderer):
    nodes, rd, _ = skull_tree
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    ref = oracle.render(nodes, rd, pos, forw, up, 1024, 1024, 2.0, 80)
    got = _gpu(skull_renderer, pos, forw, up, 1024, 1024, 2.0, 80)
    assert_same_render(got, ref, "Skull 1024 iso 80")
    _check_stats(got, ref)
    assert ref["stats"].nodes_checked == 35470121


def test_skull_sweep_properties_at_full_size(oracle, skull_vox, skull_renderer):
    """Full-size (1024^2) sweep checked through properties that need no oracle run: the OOB mask is the
    same for every iso, hit + miss + oob = pixels, gray levels stay in the shading range, every kernel
    variant agrees, and a sweep equals single renders."""
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    isos = 8
    out = skull_renderer.renderSweep(pos, forw, up, 1024, 1024, 2.0, 76, isos, taps=True)
    flags = out["cell_flag"] >> 24
    oob = flags[0] == capi.FLAG_OOB
    for k in range(isos):
        assert ((flags[k] == capi.FLAG_OOB) == oob).all()
        st = out["stats"][k]
        assert st["hit_count"] + st["miss_count"] + st["oob_count"] == 1024 * 1024
        assert st["hit_count"] == int((flags[k] == capi.FLAG_HIT).sum())
        assert st["nodes_checked"] == int(out["checked"][k].astype(np.uint64).sum())
        g = out["rgb"][k, ..., 0]
        assert ((g == 0) | (g >= 55)).all() and (g[flags[k] != capi.FLAG_HIT] == 0).all()
    single = _gpu(skull_renderer, pos, forw, up, 1024, 1024, 2.0, 80)
    assert (single["rgb"] == out["rgb"][4]).all() and (single["checked"] == out["checked"][4]).all()
    for variant in (capi.KERNEL_RESTART_LDS, capi.KERNEL_RESUME):
        skull_renderer.set_kernel(variant)
        other = _gpu(skull_renderer, pos, forw, up, 1024, 1024, 2.0, 80)
        skull_renderer.set_kernel(capi.KERNEL_DEFAULT)
        for key in ("rgb", "cell_flag", "checked", "node"):
            assert (other[key] == single[key]).all()
        assert (other["t"].view(np.uint32) == single["t"].view(np.uint32)).all()


@pytest.mark.parametrize("heuristic", ["FlexibleFirst5", "FlexibleCenterWeighted30", "FlexibleCenterWeighted75"])
def test_heuristic_trees(oracle, c60_vox, heuristic):
    k = vr.SlimKDT.newHeursiticPartition(vr.Data(c60_vox), heuristic)
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    with vr.Renderer(0) as r:
        r.upload_kdt(k)
        for iso in (30, 128, 222):
            ref = oracle.render(k.nodes, k.root_dim, pos, forw, up, 256, 256, 2.0, iso)
            got = _gpu(r, pos, forw, up, 256, 256, 2.0, iso)
            assert_same_render(got, ref, "%s iso %d" % (heuristic, iso))
            _check_stats(got, ref)
        # and the sweep kernel on the same (unbalanced) tree: a block of iso values against one oracle render each
        out = r.renderSweep(pos, forw, up, 72, 56, 2.0, 90, 24, taps=True)
        for j in range(24):
            ref = oracle.render(k.nodes, k.root_dim, pos, forw, up, 72, 56, 2.0, 90 + j)
            got = {"rgb": out["rgb"][j], "cell_flag": out["cell_flag"][j], "t": out["t"][j], "checked": out["checked"][j],
                   "node": out["node"][j], "stats": out["stats"][j]}
            assert_same_render(got, ref, "%s sweep iso %d" % (heuristic, 90 + j))
            _check_stats(got, ref)


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (37, 53), (130, 17), (256, 1)])
def test_ragged_image_sizes(oracle, c60_vox, c60_tree, c60_renderer, size):
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    w, h = size
    for variant in VARIANTS:
        c60_renderer.set_kernel(variant)
        ref = oracle.render(nodes, rd, pos, forw, up, w, h, 0.4, 100)
        got = _gpu(c60_renderer, pos, forw, up, w, h, 0.4, 100)
        assert_same_render(got, ref, "size %s variant %d" % (size, variant))
        _check_stats(got, ref)
    c60_renderer.set_kernel(capi.KERNEL_DEFAULT)


def test_small_volumes_and_odd_cameras(oracle):
    rng = np.random.default_rng(11)
    for shape, kind in [((2, 2, 2), "noise"), ((5, 9, 3), "noise"), ((33, 20, 41), "smooth"), ((40, 40, 40), "sparse")]:
        vox = random_volume(rng, shape, kind)
        k = vr.SlimKDT.newHeursiticPartition(vr.Data(vox), "AlwaysCenter")
        with vr.Renderer(0) as r:
            r.upload_kdt(k)
            for cam in [(-30.0, -300.0, -30.0), (350.0, 20.0, 10.0), (8.0, 8.0, -320.0), (-200.0, 260.0, 190.0)]:
                pos, forw, up = oracle.main_camera(shape, cam)
                for iso in (0, 40, 200):
                    ref = oracle.render(k.nodes, k.root_dim, pos, forw, up, 96, 64, 0.5, iso)
                    got = _gpu(r, pos, forw, up, 96, 64, 0.5, iso)
                    assert_same_render(got, ref, "%s %s iso %d" % (shape, cam, iso))
                    _check_stats(got, ref)


def test_axis_parallel_rays_are_oob(oracle):
    vox = np.full((8, 8, 8), 200, np.uint8)
    k = vr.SlimKDT.newHeursiticPartition(vr.Data(vox))
    pos = np.array([3.5, -300.0, 3.5], np.float32)
    forw = np.array([0.0, 1.0, 0.0], np.float32)
    up = np.array([0.0, 0.0, 1.0], np.float32)
    ref = oracle.render(k.nodes, k.root_dim, pos, forw, up, 9, 9, 0.02, 100)
    with vr.Renderer(0) as r:
        r.upload_kdt(k)
        got = _gpu(r, pos, forw, up, 9, 9, 0.02, 100)
    assert_same_render(got, ref, "axis-parallel")
    assert (got["cell_flag"][4, 4] >> 24) == capi.FLAG_OOB


def test_device_entry_point_rows_and_isos(oracle, c60_vox, c60_tree, c60_renderer):
    """vr_render_device: row ranges and iso ranges into device memory (the sharded form), assembled on the
    host, equal the full sweep."""
    import torch
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    W, H = 200, 120
    full = c60_renderer.renderSweep(pos, forw, up, W, H, 2.0, 98, 4)
    L = capi.lib()
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    pieces = []
    checked = 0
    for r0, r1 in [(0, 50), (50, 51), (51, 120)]:
        rows = r1 - r0
        d_rgb = torch.zeros((4, rows, W, 3), dtype=torch.uint8, device="cuda")
        d_stats = torch.zeros((4, 7), dtype=torch.int64, device="cuda")
        capi.check(L.vr_render_device(c60_renderer.handle, C.byref(v), 98, 4, r0, r1, d_rgb.data_ptr(),
                                      d_stats.data_ptr(), None, None), "vr_render_device")
        capi.check(L.vr_sync(c60_renderer.handle), "vr_sync")
        pieces.append(d_rgb.cpu().numpy())
        checked += int(d_stats[:, 0].sum())
    assert (np.concatenate(pieces, axis=1) == full["rgb"]).all()
    assert checked == sum(s["nodes_checked"] for s in full["stats"])


def test_calls_that_share_a_ray_list_match_the_oracle(oracle, skull_vox, skull_tree, skull_renderer):
    """The ray list of a call depends on the camera, the image size and the rows, not on the iso values: a context keeps the
    list of its last call and the next call with the same view does not make it again.  An iso slider at a fixed camera,
    with sweeps (another kind of list), other rows, another image size and another camera in between: every frame and
    every counter equals the oracle's, whatever came before."""
    import torch
    nodes, rd, _ = skull_tree
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    pos2 = (pos[0] + 7.0, pos[1] - 3.0, pos[2] + 1.0)
    L = capi.lib()

    def call(p, W, H, b, n, r0, r1, with_stats=True):
        v = vr.raytrace._view(p, forw, up, W, H, 2.0)
        d_rgb = torch.full((n, r1 - r0, W, 3), 9, dtype=torch.uint8, device="cuda")
        d_stats = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        capi.check(L.vr_render_device(skull_renderer.handle, C.byref(v), b, n, r0, r1, d_rgb.data_ptr(),
                                      d_stats.data_ptr() if with_stats else None, None, None), "vr_render_device")
        capi.check(L.vr_sync(skull_renderer.handle), "vr_sync")
        rgb, st = d_rgb.cpu().numpy(), d_stats.cpu().numpy()
        for j in (0, n - 1) if n > 1 else (0,):
            ref = oracle.render(nodes, rd, p, forw, up, W, H, 2.0, b + j)
            assert (rgb[j] == ref["rgb"][r0:r1]).all(), ("pixels", W, H, b + j, r0, r1)
            if with_stats and r0 == 0 and r1 == H:
                o = ref["stats"]
                assert st[j].tolist() == [o.nodes_checked, o.hit_count, o.miss_count, o.oob_count, o.escape_help_events, o.moves, 0], (W, H, b + j)

    W, H = 160, 120
    for b in (80, 120, 40):              # the slider: one list, three calls
        call(pos, W, H, b, 1, 0, H)
    call(pos, W, H, 30, 12, 0, H)        # a sweep: rays from the tile centres outwards, a list of its own
    call(pos, W, H, 60, 16, 0, H)        # ... kept for the next sweep
    call(pos, W, H, 90, 1, 0, H)         # back to a frame
    call(pos, W, H, 90, 3, 10, 97)       # other rows
    call(pos, W, H, 100, 1, 10, 97, with_stats=False)
    call(pos, 96, 64, 100, 1, 0, 64)     # another image size
    call(pos2, 96, 64, 100, 2, 0, 64)    # another camera, same size
    call(pos, 96, 64, 100, 2, 0, 64)
    call(pos, 512, 512, 70, 1, 0, 512)   # a larger view: the scratch grows and the list goes with the old buffer
    call(pos, 512, 512, 75, 1, 0, 512)


def test_frames_at_an_odd_address_are_cleared_whole(oracle, c60_vox, c60_tree, c60_renderer):
    """The frames are set to zero by the library's own kernel, 16 bytes per store with the odd bytes at either end one by one:
    a frame range that starts at an address that is not a multiple of 16 and has an odd length comes out as at an even one, and
    the bytes in front of and behind it are left alone."""
    import torch
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    L = capi.lib()
    W, H = 173, 91
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    for b, n in ((100, 1), (90, 3), (60, 20)):
        size = n * H * W * 3
        ref = torch.full((size,), 9, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        capi.check(L.vr_render_device(c60_renderer.handle, C.byref(v), b, n, 0, H, ref.data_ptr(), None, None, None), "vr_render_device")
        capi.check(L.vr_sync(c60_renderer.handle), "vr_sync")
        for off in (1, 5, 15):
            buf = torch.full((size + 64,), 9, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            capi.check(L.vr_render_device(c60_renderer.handle, C.byref(v), b, n, 0, H, buf.data_ptr() + off, None, None, None), "vr_render_device")
            capi.check(L.vr_sync(c60_renderer.handle), "vr_sync")
            assert bool(torch.equal(buf[off:off + size], ref)), (b, n, off)
            assert bool((buf[:off] == 9).all()) and bool((buf[off + size:] == 9).all()), (b, n, off)
    ref0 = oracle.render(c60_tree[0], c60_tree[1], pos, forw, up, W, H, 2.0, 60)
    assert (ref[:H * W * 3].cpu().numpy().reshape(H, W, 3) == ref0["rgb"]).all()


def test_error_paths_on_device(c60_tree):
    nodes, rd, _ = c60_tree
    L = capi.lib()
    with vr.Renderer(0) as r:
        v = vr.raytrace._view((0, 0, 0), (0, 1, 0), (0, 0, 1), 8, 8, 2.0)
        buf = np.zeros(8 * 8 * 3, np.uint8)
        assert L.vr_render(r.handle, C.byref(v), 0, buf.ctypes.data, None, None) == capi.VR_ESTATE
        assert L.vr_build_kdt(r.handle, 0, None) == capi.VR_ESTATE
        r.upload_kdt(vr.SlimKDT(nodes, rd))
        assert L.vr_render(r.handle, C.byref(v), 0, None, None, None) == capi.VR_EARG
        assert L.vr_render_sweep(r.handle, C.byref(v), 200, 100, buf.ctypes.data, None, None) == capi.VR_EARG
        v.width = 0
        assert L.vr_render(r.handle, C.byref(v), 0, buf.ctypes.data, None, None) == capi.VR_EARG
        assert L.vr_set_kernel(r.handle, 77) == capi.VR_EARG
        assert L.vr_upload_kdt(r.handle, nodes.ctypes.data, 0, (C.c_uint8 * 3)(1, 1, 1)) == capi.VR_EARG
        bad = nodes.copy()
        bad[0] = bad[0] & ~np.uint32(0x00FF0000)  # root partition on the box edge: never reaches a leaf
        assert L.vr_upload_kdt(r.handle, bad.ctypes.data, len(bad), (C.c_uint8 * 3)(*rd)) == capi.VR_ETREE


def test_build_on_context_and_download(oracle, c60_vox, c60_tree):
    """vr_build_kdt: the centre-split tree is built on the GPU; the downloaded array is the oracle's, word for
    word (zeros in unoccupied slots included); span heuristics go through the host builder."""
    nodes, rd, st = c60_tree
    with vr.Renderer(0) as r:
        r.upload_volume(vr.Data(c60_vox))
        bs = r.build_kdt("AlwaysCenter")
        assert bs.number_of_nodes == st.number_of_nodes and bs.layer_count == st.layer_count
        assert bs.non_center_partition_count == 0
        got, grd = r.download_kdt()
        assert (grd == rd).all() and len(got) == len(nodes) and (got == nodes).all()
        # and the tree renders like the uploaded one
        pos, forw, up = oracle.main_camera(c60_vox.shape)
        ref = oracle.render(nodes, rd, pos, forw, up, 200, 200, 2.0, 100)
        assert_same_render(_gpu(r, pos, forw, up, 200, 200, 2.0, 100), ref, "GPU-built tree")
        bs = r.build_kdt("FlexibleFirst5")
        k = vr.SlimKDT.newHeursiticPartition(vr.Data(c60_vox), "FlexibleFirst5")
        got, _ = r.download_kdt()
        assert bs.non_center_partition_count == k.build.non_center_partition_count and (got == k.nodes).all()


def test_gpu_tree_build_matches_host_builder(skull_vox, skull_tree):
    rng = np.random.default_rng(21)
    cases = [skull_vox, vr.datasets.bonsai_standin(), random_volume(rng, (2, 2, 2), "noise"),
             random_volume(rng, (3, 2, 7), "noise"), random_volume(rng, (17, 33, 9), "smooth"),
             random_volume(rng, (64, 5, 130), "sparse"), random_volume(rng, (256, 2, 2), "noise")]
    for vox in cases:
        k = vr.SlimKDT.newHeursiticPartition(vr.Data(vox), "AlwaysCenter")
        with vr.Renderer(0) as r:
            r.upload_volume(vr.Data(vox))
            bs = r.build_kdt("AlwaysCenter")
            got, grd = r.download_kdt()
        assert bs.layer_count == k.build.layer_count and bs.number_of_nodes == k.build.number_of_nodes, vox.shape
        n = min(len(got), len(k.nodes))
        assert (got[:n] == k.nodes[:n]).all() and not got[n:].any() and not k.nodes[n:].any(), vox.shape
        assert (grd == k.root_dim).all()
    assert (skull_tree[0] == vr.SlimKDT.newHeursiticPartition(vr.Data(skull_vox)).nodes[: len(skull_tree[0])]).all()


def test_standin_256_cubed_matches_oracle(oracle):
    """A 256^3 volume (the shape of the reference's largest dataset, which is not redistributable: a
    seeded synthetic stand-in): tree depth 24, 2^25 slots.  Tree built on the GPU, render checked against
    the oracle walking the host-built tree."""
    vox = vr.datasets.bonsai_standin()
    k = vr.SlimKDT.newHeursiticPartition(vr.Data(vox), "AlwaysCenter")
    pos, forw, up = oracle.main_camera(vox.shape)
    with vr.Renderer(0) as r:
        r.upload_volume(vr.Data(vox))
        r.build_kdt("AlwaysCenter")
        for iso in (40, 90, 160):
            ref = oracle.render(k.nodes, k.root_dim, pos, forw, up, 384, 384, 2.0, iso)
            got = _gpu(r, pos, forw, up, 384, 384, 2.0, iso)
            assert_same_render(got, ref, "stand-in iso %d" % iso)
            _check_stats(got, ref)
            assert ref["stats"].hit_count > 1000


def test_skull_4096_row_shards_equal_whole_frame(oracle, skull_vox, skull_tree, skull_renderer):
    """The 4096^2 configuration rendered as 4 row shards (what 4 ranks would each do) equals the frame
    rendered whole, statistics included; one 32-row band is checked against the oracle."""
    import torch
    from volumerenderer_amd import sharding
    nodes, rd, _ = skull_tree
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    W = H = 4096
    whole = skull_renderer.renderSweep(pos, forw, up, W, H, 2.0, 80, 1)
    L = capi.lib()
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    stats = np.zeros(7, np.int64)
    for rank in range(4):
        r0, r1 = sharding.row_shard(rank, 4, H)
        d_rgb = torch.zeros((1, r1 - r0, W, 3), dtype=torch.uint8, device="cuda")
        d_stats = torch.zeros((1, 7), dtype=torch.int64, device="cuda")
        capi.check(L.vr_render_device(skull_renderer.handle, C.byref(v), 80, 1, r0, r1, d_rgb.data_ptr(),
                                      d_stats.data_ptr(), None, None), "vr_render_device")
        capi.check(L.vr_sync(skull_renderer.handle), "vr_sync")
        assert (d_rgb.cpu().numpy()[0] == whole["rgb"][0, r0:r1]).all(), rank
        stats += d_stats.cpu().numpy()[0]
    ws = whole["stats"][0]
    assert stats[0] == ws["nodes_checked"] and stats[1] == ws["hit_count"] and stats[2] == ws["miss_count"]
    assert stats[3] == ws["oob_count"] and ws["hit_count"] + ws["miss_count"] + ws["oob_count"] == W * H
    band = oracle.render(nodes, rd, pos, forw, up, W, H, 2.0, 80, rows=(2040, 2072), taps=False)
    assert (band["rgb"][2040:2072] == whole["rgb"][0, 2040:2072]).all()


def test_sweep_kernel_every_iso_value_matches_oracle(oracle, c60_vox, c60_tree, c60_renderer, skull_vox, skull_tree,
                                                     skull_renderer):
    """The sweep kernel shares one traversal between all iso values that cannot change a test outcome on the
    way.  Every per-iso output -- pixels, hit cell, t bits, hit leaf, reference-defined check count, counters --
    must equal the oracle's separate render of that iso value: all 256 iso values on C60, a block on Skull,
    and iso ranges that do not start at 0 or end at 255."""
    for vox, tree, rnd, W, H, b, n in [(c60_vox, c60_tree, c60_renderer, 96, 80, 0, 256),
                                       (skull_vox, skull_tree, skull_renderer, 160, 128, 60, 40),
                                       (c60_vox, c60_tree, c60_renderer, 33, 47, 251, 5),
                                       (c60_vox, c60_tree, c60_renderer, 64, 64, 100, 1)]:
        nodes, rd, _ = tree
        pos, forw, up = oracle.main_camera(vox.shape)
        rnd.set_kernel(capi.KERNEL_SWEEP)
        out = rnd.renderSweep(pos, forw, up, W, H, 2.0, b, n, taps=True)
        rnd.set_kernel(capi.KERNEL_DEFAULT)
        for j in range(n):
            ref = oracle.render(nodes, rd, pos, forw, up, W, H, 2.0, b + j)
            got = {"rgb": out["rgb"][j], "cell_flag": out["cell_flag"][j], "t": out["t"][j], "checked": out["checked"][j],
                   "node": out["node"][j], "stats": out["stats"][j]}
            assert_same_render(got, ref, "sweep kernel iso %d" % (b + j))
            _check_stats(got, ref)


def test_sweep_kernel_equals_per_iso_kernel_at_full_size(oracle, skull_vox, skull_renderer):
    """Skull 1024^2, all 256 iso values: frames and counters of the sweep kernel and of the one-traversal-per-iso
    kernel are identical."""
    import torch
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    L = capi.lib()
    W = H = 1024
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    res = []
    for variant in (capi.KERNEL_PIPELINE, capi.KERNEL_SWEEP):
        skull_renderer.set_kernel(variant)
        d_rgb = torch.empty((256, H, W, 3), dtype=torch.uint8, device="cuda")
        d_stats = torch.zeros((256, 7), dtype=torch.int64, device="cuda")
        capi.check(L.vr_render_device(skull_renderer.handle, C.byref(v), 0, 256, 0, H, d_rgb.data_ptr(), d_stats.data_ptr(),
                                      None, None), "vr_render_device")
        capi.check(L.vr_sync(skull_renderer.handle), "vr_sync")
        res.append((d_rgb, d_stats.cpu().numpy()))
    skull_renderer.set_kernel(capi.KERNEL_DEFAULT)
    assert (res[0][1] == res[1][1]).all()
    # every frame's counters are the oracle's (tests/golden/fullsize_oracle.json, made by the CPU oracle)
    assert (res[0][1] == helpers.fullsize_golden_stats("skull_1024")).all() and int(res[0][1][:, 6].sum()) == 0
    assert bool(torch.equal(res[0][0], res[1][0]))


def test_sweep_kernel_with_no_ray_entering_the_volume(oracle, c60_vox, c60_renderer):
    """A camera that looks away from the volume: every frame black, every ray OOB in every iso value, nothing hangs."""
    pos = np.array([-30.0, -300.0, -30.0], np.float32)
    forw = np.array([0.0, -1.0, 0.0], np.float32)
    up = np.array([0.0, 0.0, 1.0], np.float32)
    for variant in (capi.KERNEL_SWEEP, capi.KERNEL_PIPELINE):
        c60_renderer.set_kernel(variant)
        out = c60_renderer.renderSweep(pos, forw, up, 70, 50, 2.0, 0, 256)
        c60_renderer.set_kernel(capi.KERNEL_DEFAULT)
        assert not out["rgb"].any()
        assert all(s["oob_count"] == 70 * 50 and s["nodes_checked"] == 0 and s["hit_count"] == 0 for s in out["stats"])


def test_page_locked_output_buffer(oracle, c60_vox, c60_tree, c60_renderer):
    """vr_alloc_host: frames delivered into page-locked memory equal frames delivered into ordinary memory."""
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    buf = c60_renderer.pinned_frames(6, 40, 64)
    a = c60_renderer.renderSweep(pos, forw, up, 64, 40, 2.0, 97, 6, out=buf)
    b = c60_renderer.renderSweep(pos, forw, up, 64, 40, 2.0, 97, 6)
    assert a["rgb"] is buf and (buf == b["rgb"]).all() and buf.any()


def test_page_locked_chunked_sweep_equals_pageable_and_reference_renders(oracle, c60_vox, c60_renderer, golden_gray):
    """vr_render_sweep into page-locked memory goes in chunks of 64 iso values whose copies overlap the next
    chunk's render (from 128 iso values and 32 MiB per chunk on).  C60 512^2: all 256 frames equal the
    reference's renders and the pageable call (frames and counters); 130 iso values leave a 2-frame tail chunk,
    which runs the one-traversal-per-iso kernel."""
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    for b, n in [(0, 256), (60, 130)]:
        buf = c60_renderer.pinned_frames(n, 512, 512)
        buf[...] = 7
        a = c60_renderer.renderSweep(pos, forw, up, 512, 512, 2.0, b, n, out=buf)
        p = c60_renderer.renderSweep(pos, forw, up, 512, 512, 2.0, b, n)
        assert (a["rgb"][..., 0] == golden_gray[b:b + n]).all() and (a["rgb"] == p["rgb"]).all()
        assert a["stats"] == p["stats"] and all(s["errors"] == 0 for s in a["stats"])
        del a, buf


def test_standin_256_cubed_sweep_kernel_every_iso_value_matches_oracle(oracle, standin_tree, standin_renderer):
    """The 0..255 sweep of the 256^3 volume (depth-24 tree: 25 path-stack rows, 13 tree levels in LDS) through the
    sweep kernel: every iso value against its own oracle render, all taps and counters."""
    vox, k = standin_tree
    pos, forw, up = oracle.main_camera(vox.shape)
    standin_renderer.set_kernel(capi.KERNEL_SWEEP)
    out = standin_renderer.renderSweep(pos, forw, up, 96, 80, 2.0, 0, 256, taps=True)
    standin_renderer.set_kernel(capi.KERNEL_DEFAULT)
    hits = 0
    for j in range(256):
        ref = oracle.render(k.nodes, k.root_dim, pos, forw, up, 96, 80, 2.0, j)
        got = {"rgb": out["rgb"][j], "cell_flag": out["cell_flag"][j], "t": out["t"][j], "checked": out["checked"][j],
               "node": out["node"][j], "stats": out["stats"][j]}
        assert_same_render(got, ref, "stand-in sweep kernel iso %d" % j)
        _check_stats(got, ref)
        hits += ref["stats"].hit_count
    assert hits > 100000


def test_standin_256_cubed_full_size_sweep_equals_per_iso_kernel_and_iso_shards(oracle, standin_tree, standin_renderer):
    """The 256^3 volume at 1024^2, iso 0..255: frames and all 7 counters of the sweep kernel equal those of the
    one-traversal-per-iso kernel, and the 8 iso shards of an 8-GPU run, rendered one after the other, equal the
    whole sweep."""
    import torch
    from volumerenderer_amd import sharding
    vox, _ = standin_tree
    pos, forw, up = oracle.main_camera(vox.shape)
    L = capi.lib()
    W = H = 1024
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)

    def run(b, n):
        d_rgb = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
        d_stats = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
        capi.check(L.vr_render_device(standin_renderer.handle, C.byref(v), b, n, 0, H, d_rgb.data_ptr(), d_stats.data_ptr(),
                                      None, None), "vr_render_device")
        capi.check(L.vr_sync(standin_renderer.handle), "vr_sync")
        return d_rgb, d_stats.cpu().numpy()

    standin_renderer.set_kernel(capi.KERNEL_PIPELINE)
    per_iso = run(0, 256)
    standin_renderer.set_kernel(capi.KERNEL_SWEEP)
    sweep = run(0, 256)
    standin_renderer.set_kernel(capi.KERNEL_DEFAULT)
    assert (per_iso[1] == sweep[1]).all() and int(sweep[1][:, 6].sum()) == 0 and int(sweep[1][:, 1].sum()) > 10 ** 7
    assert bool(torch.equal(per_iso[0], sweep[0]))
    del per_iso
    for rank in range(8):
        b, n = sharding.iso_shard(rank, 8)
        part = run(b, n)
        assert (part[1] == sweep[1][b:b + n]).all(), rank
        assert bool(torch.equal(part[0], sweep[0][b:b + n])), rank


def test_sweep_kernel_on_random_small_volumes(oracle):
    """Noise, smooth and sparse volumes of odd shapes from odd camera positions, all 256 iso values through the
    sweep kernel (noise gives almost every iso value its own walk, sparse volumes give long shared ones)."""
    rng = np.random.default_rng(23)
    for shape, kind, cam in [((5, 9, 3), "noise", (-30.0, -300.0, -30.0)), ((20, 33, 17), "noise", (350.0, 20.0, 10.0)),
                             ((33, 20, 41), "smooth", (8.0, 8.0, -320.0)), ((40, 40, 40), "sparse", (-200.0, 260.0, 190.0)),
                             ((2, 2, 2), "noise", (-30.0, -300.0, -30.0))]:
        vox = random_volume(rng, shape, kind)
        k = vr.SlimKDT.newHeursiticPartition(vr.Data(vox), "AlwaysCenter")
        pos, forw, up = oracle.main_camera(shape, cam)
        with vr.Renderer(0) as r:
            r.upload_kdt(k)
            r.set_kernel(capi.KERNEL_SWEEP)
            out = r.renderSweep(pos, forw, up, 48, 40, 0.5, 0, 256, taps=True)
        for j in range(0, 256, 3):
            ref = oracle.render(k.nodes, k.root_dim, pos, forw, up, 48, 40, 0.5, j)
            got = {"rgb": out["rgb"][j], "cell_flag": out["cell_flag"][j], "t": out["t"][j], "checked": out["checked"][j],
                   "node": out["node"][j], "stats": out["stats"][j]}
            assert_same_render(got, ref, "%s %s iso %d" % (shape, kind, j))
            _check_stats(got, ref)


def test_vr_bench_driver_reproduces_reference_renders(tmp_path, golden_gray):
    """The C++ driver with the shape of the reference's main: writes renders/out_iso{d}.png and prints the
    comparison table; its files decode to the reference's own images."""
    import os
    import subprocess
    PIL = pytest.importorskip("PIL.Image")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "volumerenderer_amd", "vr_bench")
    for extra, lo, hi in [([], 97, 103), (["--sweep"], 90, 110), (["--gpus", "3", "--shard", "rows"], 95, 102),
                          (["--gpus", "2"], 99, 104), (["--gpu-build", "--sweep"], 100, 103)]:
        out = subprocess.run([exe, "--volume", os.path.join(root, "tests", "data", "C60.vol"), "--size", "512", "--iso",
                              "%d:%d" % (lo, hi), "--heuristics", "center", "--out", str(tmp_path)] + extra,
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert "(ms/frame)" in out.stdout and "nodes: 500093" in out.stdout and "layers: 18" in out.stdout
        for iso in range(lo, hi):
            img = np.asarray(PIL.open(tmp_path / ("out_iso%d.png" % iso)).convert("RGB"))
            assert (img[..., 0] == golden_gray[iso]).all() and (img[..., 1] == img[..., 0]).all()
            os.remove(tmp_path / ("out_iso%d.png" % iso))


@pytest.mark.parametrize("shard", ["iso", "rows"])
def test_one_thread_several_contexts_equal_one_context(oracle, c60_vox, c60_tree, c60_renderer, shard):
    """vr_mgpu_render_sweep: the iso loop split over three contexts by one host thread (all on device 0 here; one
    per GPU on a node) leaves frames and per-iso counters exactly as one context does -- uneven splits, more
    contexts than frames, page-locked and pageable output."""
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    extra = [vr.Renderer(0) for _ in range(2)]
    try:
        for r in extra:
            r.upload_kdt(vr.SlimKDT(nodes, rd))
        group = [c60_renderer] + extra
        for W, H, b, n, pinned in [(200, 131, 90, 7, False), (64, 50, 100, 2, True), (96, 2, 0, 256, False)]:
            one = c60_renderer.renderSweep(pos, forw, up, W, H, 2.0, b, n)
            out = c60_renderer.pinned_frames(n, H, W) if pinned else None
            got = vr.raytrace.renderSweepMulti(group, pos, forw, up, W, H, 2.0, b, n, shard=shard, out=out)
            assert (got["rgb"] == one["rgb"]).all(), (W, H, b, n)
            assert got["stats"] == one["stats"], (W, H, b, n)
        L = capi.lib()
        v = vr.raytrace._view(pos, forw, up, 8, 8, 2.0)
        buf = np.zeros((1, 8, 8, 3), np.uint8)
        twice = (C.c_void_p * 2)(c60_renderer.handle, c60_renderer.handle)
        assert L.vr_mgpu_render_sweep(twice, 2, C.byref(v), 0, 1, 0, buf.ctypes.data, None) == capi.VR_EARG
        assert L.vr_mgpu_render_sweep(twice, 1, C.byref(v), 0, 1, 2, buf.ctypes.data, None) == capi.VR_EARG
    finally:
        for r in extra:
            r.close()


def test_bench_line_has_the_contract_fields():
    """bench.py prints one JSON line; the keys the harness reads must all be there (small image, 2 steps)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--width", "256"],
                         capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [x for x in out.stdout.splitlines() if x.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "Mrays/s" and d["errors"] == 0
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert d["roofline"]["bound"] in ("hbm", "latency") and d["roofline"]["peak"] == 8000.0
    # the floors a step can be held against: the frames written once at the HBM peak, and the sweep's longest single walk
    assert 0 < d["roofline"]["hbm_floor_ms"] < d["ms_per_step"] and d["roofline"]["step_over_hbm_floor"] > 1
    assert 0 < d["roofline"]["latency_floor_ms"] <= d["roofline"]["kernel_ms"] * 1.05 and d["roofline"]["latency_floor"]["checks"] > 100
    assert d["roofline"]["bound"] == ("latency" if d["roofline"]["latency_floor_ms"] > d["roofline"]["hbm_floor_ms"] else "hbm")
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / 8000.0) < 1e-9
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port"
    assert d["value"] > 100 * d["cpu_baseline"]["value"]
    # the timed steps repeat one call, so they run with the tiles longest first; what a caller gets who renders a view
    # once is reported next to it
    assert d["roofline"]["tiles_longest_first"] is True
    assert d["first_call"]["ms_per_step"] > 0 and d["first_call"]["kernel_ms"] > 0 and d["first_call"]["value"] > 0
    assert d["first_call"]["repeated_after_another_call_ms"] > 0


SPAN_HEURISTICS = ["MinimizeSpan", "FlexibleFirst5", "FlexibleFirst10", "FlexibleAfter10", "FlexibleAfter15",
                   "FlexibleAfter20", "FlexibleCenterWeighted5", "FlexibleCenterWeighted10", "FlexibleCenterWeighted30",
                   "FlexibleCenterWeighted50", "FlexibleCenterWeighted75", "FlexibleCenterWeighted100"]


def test_gpu_span_heuristic_trees_match_host_builder_and_oracle(oracle, c60_vox, skull_vox):
    """The span heuristics built on the GPU level by level: the node array, the statistics and the failure case are the
    oracle's (oracle.build_kdt, run here for every case but Skull, whose trees take it minutes: those are compared with the
    product's host builder here and with the oracle's committed fixture in tests/test_gpu_golden_fullsize.py), word for word."""
    rng = np.random.default_rng(33)
    cases = [(c60_vox[20:40, 18:40, 20:38], SPAN_HEURISTICS), (random_volume(rng, (9, 30, 5), "noise"), SPAN_HEURISTICS),
             (random_volume(rng, (40, 37, 44), "smooth"), SPAN_HEURISTICS[1:]), (random_volume(rng, (2, 2, 2), "noise"), SPAN_HEURISTICS[:3]),
             (c60_vox, ["FlexibleFirst5", "FlexibleAfter15", "FlexibleCenterWeighted50", "FlexibleCenterWeighted100"]),
             (skull_vox, ["FlexibleFirst5", "FlexibleCenterWeighted100"])]
    built = 0
    for vox, heuristics in cases:
        vox = np.ascontiguousarray(vox)
        with vr.Renderer(0) as r:
            r.upload_volume(vr.Data(vox))
            for h in heuristics:
                try:
                    k = vr.SlimKDT.newHeursiticPartition(vr.Data(vox), h)
                except capi.VrError as e:
                    assert e.code == capi.VR_ETOODEEP
                    with pytest.raises(capi.VrError) as g:
                        r.build_kdt(h)
                    assert g.value.code == capi.VR_ETOODEEP, (vox.shape, h)
                    continue
                bs = r.build_kdt(h)
                got, grd = r.download_kdt()
                assert (bs.layer_count, bs.number_of_nodes, bs.non_center_partition_count) == (
                    k.build.layer_count, k.build.number_of_nodes, k.build.non_center_partition_count), (vox.shape, h)
                n = min(len(got), len(k.nodes))
                assert (got[:n] == k.nodes[:n]).all() and not got[n:].any() and not k.nodes[n:].any(), (vox.shape, h)
                assert (grd == k.root_dim).all()
                if vox.size < 1 << 20:  # the oracle's own tree, built here
                    o_nodes, o_rd, o_st = oracle.build_kdt(vox, h)
                    n = min(len(got), len(o_nodes))
                    assert (got[:n] == o_nodes[:n]).all() and not got[n:].any() and not o_nodes[n:].any(), (vox.shape, h, "oracle")
                    assert (grd == o_rd).all() and (bs.layer_count, bs.number_of_nodes, bs.non_center_partition_count) == (
                        o_st.layer_count, o_st.number_of_nodes, o_st.non_center_partition_count), (vox.shape, h, "oracle")
                built += 1
            # the tree built last renders like the oracle walking the host-built one
            pos, forw, up = oracle.main_camera(vox.shape)
            ref = oracle.render(k.nodes, k.root_dim, pos, forw, up, 96, 96, 2.0, 100)
            assert_same_render(_gpu(r, pos, forw, up, 96, 96, 2.0, 100), ref, "GPU-built %s tree" % h)
    assert built >= 25


def test_band_entry_point_equals_whole_frames(oracle, skull_vox, skull_renderer):
    """vr_render_device_bands: the rows of a sweep dealt out in bands to 3 ranks (uneven: 200 rows = 7 bands of 32,
    the last one short), every rank's packed output put back in place, equals the sweep rendered whole -- frames,
    taps and the sum of the per-iso counters; also bands that are not a multiple of the tile height."""
    import torch
    from volumerenderer_amd import sharding
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    L = capi.lib()
    for W, H, band, world, b, n in [(256, 200, 32, 3, 70, 12), (96, 61, 5, 4, 80, 1), (64, 64, 32, 8, 0, 9)]:
        v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
        whole = skull_renderer.renderSweep(pos, forw, up, W, H, 2.0, b, n, taps=True)
        rgb = np.zeros((n, H, W, 3), np.uint8)
        checked = np.zeros((n, H, W), np.uint32)
        stats = np.zeros((n, 7), np.int64)
        for rank in range(world):
            sh = sharding.band_shard(rank, world, H, band)
            if sh["n_bands"] == 0:
                continue
            rows = sharding.band_rows_of(rank, world, H, band)
            d_rgb = torch.full((n, sh["n_rows"], W, 3), 3, dtype=torch.uint8, device="cuda")
            d_chk = torch.zeros((n, sh["n_rows"], W), dtype=torch.int32, device="cuda")
            d_stats = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
            taps = capi.Taps(None, None, d_chk.data_ptr(), None)
            capi.check(L.vr_render_device_bands(skull_renderer.handle, C.byref(v), b, n, sh["row_begin"], sh["band_rows"],
                                                sh["band_pitch"], sh["n_bands"], d_rgb.data_ptr(), d_stats.data_ptr(),
                                                C.byref(taps), None), "vr_render_device_bands")
            capi.check(L.vr_sync(skull_renderer.handle), "vr_sync")
            rgb[:, rows] = d_rgb.cpu().numpy()
            checked[:, rows] = d_chk.cpu().numpy().view(np.uint32)
            stats += d_stats.cpu().numpy()
        assert (rgb == whole["rgb"]).all() and (checked == whole["checked"]).all(), (W, H, band, world)
        for k in range(n):
            ws = whole["stats"][k]
            assert list(stats[k]) == [ws[f] for f in ("nodes_checked", "hit_count", "miss_count", "oob_count",
                                                       "escape_help_events", "moves", "errors")]
    v = vr.raytrace._view(pos, forw, up, 64, 64, 2.0)
    d = torch.zeros(64 * 64 * 3, dtype=torch.uint8, device="cuda")
    assert L.vr_render_device_bands(skull_renderer.handle, C.byref(v), 0, 1, 40, 32, 16, 2, d.data_ptr(), None, None, None) == capi.VR_EARG
    assert L.vr_render_device_bands(skull_renderer.handle, C.byref(v), 0, 1, 0, 32, 32, 3, d.data_ptr(), None, None, None) == capi.VR_EARG


def _two_rank_worker(rank, world, port, tmp):
    """One of two gloo ranks that share GPU 0: renders its band shard and its iso shard of a Skull sweep with the
    product (vr_render_device_bands / vr_render_device on a stream of its own), gathers to rank 0."""
    import os
    import sys
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from volumerenderer_amd import sharding
    data = vr.Data.loadUCD(os.path.join(root, "tests", "data", "Skull.vol.gz"))
    pos, forw, up = vr.main_camera(data.resulution)
    W, H, b, n = 320, 200, 72, 16
    L = capi.lib()
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    with vr.Renderer(0) as r, torch.cuda.stream(stream):
        r.upload_volume(data)
        r.build_kdt("AlwaysCenter")
        # rows in bands, all iso values on every rank
        sh = sharding.band_shard(rank, world, H)
        d_rgb = torch.empty((n, sh["n_rows"], W, 3), dtype=torch.uint8, device="cuda")
        d_stats = torch.zeros((n, 7), dtype=torch.int64, device="cuda")
        capi.check(L.vr_render_device_bands(r.handle, C.byref(v), b, n, sh["row_begin"], sh["band_rows"], sh["band_pitch"],
                                            sh["n_bands"], d_rgb.data_ptr(), d_stats.data_ptr(), None,
                                            C.c_void_p(stream.cuda_stream)), "vr_render_device_bands")
        rows = [sharding.band_rows_of(k, world, H) for k in range(world)]
        banded = sharding.gather_rows(d_rgb.cpu(), rows, H, dst=0)  # .cpu() waits on the stream the render is on
        st_rows = d_stats.cpu()
        dist.reduce(st_rows, dst=0)
        # iso values in blocks, whole frames on every rank
        ib, ic = sharding.iso_shard(rank, world, b, n)
        d_rgb2 = torch.empty((ic, H, W, 3), dtype=torch.uint8, device="cuda")
        d_stats2 = torch.zeros((ic, 7), dtype=torch.int64, device="cuda")
        capi.check(L.vr_render_device(r.handle, C.byref(v), ib, ic, 0, H, d_rgb2.data_ptr(), d_stats2.data_ptr(), None,
                                      C.c_void_p(stream.cuda_stream)), "vr_render_device")
        counts = [sharding.iso_shard(k, world, b, n)[1] for k in range(world)]
        by_iso = sharding.gather_frames(d_rgb2.cpu(), counts, dst=0)
        st_iso = sharding.gather_frames(d_stats2.cpu(), counts, dst=0)
        if rank == 0:
            whole = r.renderSweep(pos, forw, up, W, H, 2.0, b, n)
            ref_stats = np.array([[s[f] for f in ("nodes_checked", "hit_count", "miss_count", "oob_count",
                                                  "escape_help_events", "moves", "errors")] for s in whole["stats"]])
            assert (banded.numpy() == whole["rgb"]).all() and (by_iso.numpy() == whole["rgb"]).all()
            assert (st_rows.numpy() == ref_stats).all() and (st_iso.numpy() == ref_stats).all()
            assert whole["rgb"].any()
            open(os.path.join(tmp, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_render_their_shards_with_the_product(tmp_path):
    """The product under a process group: two ranks (sharing GPU 0 here, one GPU each on a node) render band shards
    and iso shards of a sweep through the C ABI and gather them; rank 0 compares with the sweep rendered whole."""
    import os
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    mp.get_context("spawn")
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()

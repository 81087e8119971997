"""Volumes beyond the reference's u8 coordinates (up to 1024 voxels per axis), on the CPU.

No reference artefact exists for such volumes (SPACESIZE = u8, src/typedef.zig:1).  The wide oracle is
the same C file built with u16 coordinates; these tests tie it to the pinned u8 oracle wherever both
apply, and tie the product's wide lane code (vr_trace_wide.h, run on the CPU) to the wide oracle.
"""
import numpy as np
import pytest

from helpers import assert_same_render, emu_union_wide, emu_wide, random_volume


def narrow_to_wide_words(nodes):
    """SlimNode word (partition 8 bits, axis at bit 24) -> wide node word (partition 12 bits, axis at bit 28)."""
    nodes = nodes.astype(np.uint32)
    return (nodes & np.uint32(0x00ffffff)) | (((nodes >> np.uint32(24)) & np.uint32(3)) << np.uint32(28))


def narrow_to_wide_cell_flag(cf):
    x, y, z, flag = cf & 0xff, (cf >> 8) & 0xff, (cf >> 16) & 0xff, cf >> 24
    return (x | (y << 10) | (z << 20) | (flag << 30)).astype(np.uint32)


@pytest.mark.parametrize("heuristic", ["AlwaysCenter", "FlexibleFirst5", "FlexibleCenterWeighted100"])
def test_wide_oracle_equals_u8_oracle_on_c60(oracle, c60_vox, heuristic):
    vox = c60_vox if heuristic == "AlwaysCenter" else c60_vox[20:40, 18:40, 20:38]
    n8, rd8, st8 = oracle.build_kdt(vox, heuristic)
    nw, rdw, stw = oracle.build_kdt(vox, heuristic, wide=True)
    assert (rd8 == rdw).all() and len(n8) == len(nw)
    assert (narrow_to_wide_words(n8) == nw).all()
    assert (st8.number_of_nodes, st8.layer_count, st8.non_center_partition_count) == (
        stw.number_of_nodes, stw.layer_count, stw.non_center_partition_count)
    pos, forw, up = oracle.main_camera(vox.shape)
    for iso in (0, 60, 100, 255):
        a = oracle.render(n8, rd8, pos, forw, up, 96, 96, 2.0, iso)
        b = oracle.render(nw, rdw, pos, forw, up, 96, 96, 2.0, iso, wide=True)
        a["cell_flag"] = narrow_to_wide_cell_flag(a["cell_flag"])
        assert_same_render(b, a, "wide oracle vs u8 oracle, %s iso %d" % (heuristic, iso))
        assert a["stats"].nodes_checked == b["stats"].nodes_checked


def test_wide_oracle_reproduces_the_reference_renders(oracle, c60_vox, golden_gray):
    """The wide build against the reference's own renders/out_iso{d}.png (C60, 512 x 512)."""
    nw, rdw, _ = oracle.build_kdt(c60_vox, wide=True)
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    for iso in (0, 57, 100, 140, 255):
        got = oracle.render(nw, rdw, pos, forw, up, 512, 512, 2.0, iso, wide=True, taps=False)
        assert (got["rgb"][:, :, 0] == golden_gray[iso]).all()


WIDE_SHAPES = [((300, 9, 7), "noise"), ((5, 700, 6), "smooth"), ((4, 6, 1024), "sparse"), ((260, 4, 258), "noise")]


@pytest.mark.parametrize("shape,kind", WIDE_SHAPES)
def test_wide_lane_code_matches_wide_oracle(oracle, emu_lib, shape, kind):
    rng = np.random.default_rng(sum(shape))
    vox = random_volume(rng, shape, kind)
    nodes, rd, st = oracle.build_kdt(vox, wide=True)
    assert int(rd.max()) > 255
    cams = [(-30.0, -300.0, -30.0), (float(shape[0]) + 40.0, 20.0, 10.0), (8.0, 8.0, -320.0)]
    hits = 0
    for pos in cams:
        pos, forw, up = oracle.main_camera(shape, pos)
        for vp in (0.2, 3.0):
            for iso in (0, 90, 200):
                ref = oracle.render(nodes, rd, pos, forw, up, 48, 40, vp, iso, wide=True)
                for pop_trips, ahead_trips in ((0, 0), (1, 0), (3, 0), (3, 6), (0, 1), (0, 2)):
                    got = emu_wide(emu_lib, nodes, rd, pos, forw, up, 48, 40, vp, iso, pop_trips, ahead_trips)
                    assert_same_render(got, ref, "%s %s iso %d pop bound %d, reading ahead %d" % (shape, pos, iso, pop_trips, ahead_trips))
                    assert int(got["sums"][0]) == ref["stats"].nodes_checked and int(got["sums"][6]) == 0
                hits += ref["stats"].hit_count
    assert hits > 0


def test_wide_lane_code_on_c60_equals_u8_oracle(oracle, emu_lib, c60_vox, c60_tree):
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    for iso in (30, 100, 180):
        ref = oracle.render(nodes, rd, pos, forw, up, 128, 128, 2.0, iso)
        ref["cell_flag"] = narrow_to_wide_cell_flag(ref["cell_flag"])
        got = emu_wide(emu_lib, narrow_to_wide_words(nodes), rd.astype(np.uint16), pos, forw, up, 128, 128, 2.0, iso)
        assert_same_render(got, ref, "wide lane code on C60 iso %d" % iso)


def check_wide(nodes, rd):
    import ctypes as C
    from volumerenderer_amd import capi
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    depth, count = C.c_uint32(), C.c_uint64()
    rc = capi.lib().vr_host_check_kdt_wide(nodes.ctypes.data, len(nodes), (C.c_uint16 * 3)(*[int(x) for x in rd]),
                                           C.byref(depth), C.byref(count))
    return rc, depth.value, count.value


def test_wide_tree_validation(oracle):
    """The kernels walk wide trees without per-node checks; the validating walk in front of the upload must
    accept what the builder produces and refuse everything a descent could leave the array or loop on."""
    from volumerenderer_amd import capi
    rng = np.random.default_rng(5)
    vox = random_volume(rng, (300, 9, 7), "noise")
    nodes, rd, st = oracle.build_kdt(vox, wide=True)
    rc, depth, count = check_wide(nodes, rd)
    assert (rc, depth, count) == (0, st.layer_count, st.number_of_nodes)
    assert check_wide(nodes[:1000], rd)[0] == capi.VR_ETREE                      # children beyond the array
    bad = nodes.copy(); bad[0] = (bad[0] & 0x0000ffff) | (299 << 16)             # partition on the box face
    assert check_wide(bad, rd)[0] == capi.VR_ETREE
    bad = nodes.copy(); bad[5] |= 3 << 28                                        # axis 3
    assert check_wide(bad, rd)[0] == capi.VR_ETREE
    junk = rng.integers(0, 2 ** 32, 4096, dtype=np.uint64).astype(np.uint32)
    assert check_wide(junk, [300, 200, 100])[0] == capi.VR_ETREE
    assert check_wide(nodes, [1024, 8, 6])[0] == capi.VR_EARG


@pytest.mark.parametrize("shape,kind", [((300, 9, 11), "noise"), ((10, 280, 12), "smooth"), ((12, 10, 270), "sparse")])
def test_wide_union_walk_equals_one_render_per_iso(oracle, emu_lib, shape, kind):
    """ONE traversal per ray for a whole iso range on wide trees: every per-iso output and counter equals the wide oracle's
    separate render of that iso value."""
    rng = np.random.default_rng(sum(shape) + 1)
    vox = random_volume(rng, shape, kind)
    nodes, rd, st = oracle.build_kdt(vox, wide=True)
    for cam, vp in [((-30.0, -300.0, -30.0), 0.2), ((float(shape[0]) + 40.0, 20.0, 10.0), 3.0), ((8.0, 8.0, -320.0), 0.2)]:
        pos, forw, up = oracle.main_camera(shape, cam)
        for b, n, trips in [(0, 256, 5), (40, 100, 1), (250, 6, 64)]:
            got = emu_union_wide(emu_lib, nodes, rd, pos, forw, up, 32, 24, vp, b, n, max_trips=trips)
            for j in range(0, n, 1 if n < 50 else 7):
                ref = oracle.render(nodes, rd, pos, forw, up, 32, 24, vp, b + j, wide=True)
                hit = (ref["cell_flag"] >> 30) == 1
                assert (got["gray"][j] == ref["rgb"][..., 0]).all(), (shape, cam, b, j)
                assert (got["cell_flag"][j] == ref["cell_flag"]).all() and (got["node"][j] == ref["node"]).all(), (shape, cam, b, j)
                assert (got["t"][j].view(np.uint32)[hit] == ref["t"].view(np.uint32)[hit]).all(), (shape, cam, b, j)
                s = ref["stats"]
                assert tuple(int(x) for x in got["sums"][j][:6]) == (s.nodes_checked, s.hit_count, s.miss_count, s.oob_count,
                                                                     s.escape_help_events, s.moves), (shape, cam, b, j)

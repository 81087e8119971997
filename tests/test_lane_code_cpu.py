"""The kernels' lane code (vr_trace_core.h), compiled for the host, against the oracle.  This checks
the algorithmic claim of the resume traversal -- same hit, same t bits, same reference-defined
checked count as the root-restart loop -- before anything runs on a GPU."""
import numpy as np
import pytest

from helpers import assert_same_render, emu_render, random_volume


@pytest.mark.parametrize("variant", [1, 3, 101, 103, 106])
@pytest.mark.parametrize("iso", [0, 1, 57, 100, 200, 255])
def test_c60_lane_code(oracle, emu_lib, c60_vox, c60_tree, variant, iso):
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    ref = oracle.render(nodes, rd, pos, forw, up, 160, 160, 2.0, iso)
    got = emu_render(emu_lib, variant, nodes, rd, pos, forw, up, 160, 160, 2.0, iso)
    assert_same_render(got, ref, "C60 iso %d variant %d" % (iso, variant))
    assert int(got["sums"][0]) == ref["stats"].nodes_checked
    assert int(got["sums"][6]) == 0


@pytest.mark.parametrize("variant", [1, 3])
def test_skull_lane_code(oracle, emu_lib, skull_vox, skull_tree, variant):
    nodes, rd, _ = skull_tree
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    for iso in (0, 80, 255):
        ref = oracle.render(nodes, rd, pos, forw, up, 128, 128, 2.0, iso)
        got = emu_render(emu_lib, variant, nodes, rd, pos, forw, up, 128, 128, 2.0, iso)
        assert_same_render(got, ref, "Skull iso %d variant %d" % (iso, variant))


@pytest.mark.parametrize("heuristic", ["FlexibleFirst5", "FlexibleCenterWeighted30"])
def test_heuristic_trees_lane_code(oracle, emu_lib, c60_vox, heuristic):
    nodes, rd, _ = oracle.build_kdt(c60_vox, heuristic)
    nodes = oracle.trim_nodes(nodes)
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    for iso in (30, 128):
        ref = oracle.render(nodes, rd, pos, forw, up, 96, 96, 2.0, iso)
        for variant in (3,):
            got = emu_render(emu_lib, variant, nodes, rd, pos, forw, up, 96, 96, 2.0, iso)
            assert_same_render(got, ref, "%s iso %d v%d" % (heuristic, iso, variant))


def test_odd_cameras_and_small_volumes(oracle, emu_lib):
    rng = np.random.default_rng(7)
    for shape, kind in [((5, 9, 3), "noise"), ((33, 20, 41), "smooth"), ((16, 16, 16), "sparse"), ((2, 2, 2), "noise")]:
        vox = random_volume(rng, shape, kind)
        nodes, rd, _ = oracle.build_kdt(vox)
        nodes = oracle.trim_nodes(nodes)
        cams = [(-30.0, -300.0, -30.0), (350.0, 20.0, 10.0), (8.0, 8.0, -320.0), (-200.0, 260.0, 190.0)]
        for pos in cams:
            pos, forw, up = oracle.main_camera(shape, pos)
            for iso in (0, 40, 200):
                ref = oracle.render(nodes, rd, pos, forw, up, 64, 48, 0.5, iso)
                for variant in (1, 3, 102, 105):
                    got = emu_render(emu_lib, variant, nodes, rd, pos, forw, up, 64, 48, 0.5, iso)
                    assert_same_render(got, ref, "%s %s iso %d v%d" % (shape, pos, iso, variant))


def test_axis_parallel_rays_are_oob(oracle, emu_lib):
    # |d_a| < 1e-8 makes the entry plane hit +inf, so the ray is reported out of bounds (src/geo.zig:417)
    vox = np.full((8, 8, 8), 200, np.uint8)
    nodes, rd, _ = oracle.build_kdt(vox)
    nodes = oracle.trim_nodes(nodes)
    pos = np.array([3.5, -300.0, 3.5], np.float32)
    forw = np.array([0.0, 1.0, 0.0], np.float32)
    up = np.array([0.0, 0.0, 1.0], np.float32)
    ref = oracle.render(nodes, rd, pos, forw, up, 9, 9, 0.02, 100)
    got = emu_render(emu_lib, 3, nodes, rd, pos, forw, up, 9, 9, 0.02, 100)
    assert_same_render(got, ref, "axis-parallel")
    assert (ref["cell_flag"][4, 4] >> 24) == 2  # the centre ray is exactly axis-parallel


@pytest.mark.parametrize("chunk,share,ahead", [(0, 0, 0), (32, 0, 0), (7, 0, 1), (0, 2, 0), (32, 2, 1), (7, 1, 0), (0, 8, 1)])
def test_shared_traversals_across_iso_values_equal_one_render_per_iso(oracle, emu_lib, c60_vox, c60_tree, chunk, share, ahead):
    """The sweep kernel traces a ray once per interval of iso values that cannot change any test outcome;
    every per-iso output must still equal the oracle's separate render of that iso value, bit for bit.  With
    `share` > 0 the walks of a ray also carry on from kept states of its earlier walks instead of the entry point."""
    from helpers import emu_sweep
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    W = H = 40
    got = emu_sweep(emu_lib, nodes, rd, pos, forw, up, W, H, 2.0, 0, 256, chunk=chunk, max_trips=6, share=share, ahead=ahead)
    n_active = 0
    for iso in range(256):
        ref = oracle.render(nodes, rd, pos, forw, up, W, H, 2.0, iso)
        assert (got["gray"][iso] == ref["rgb"][..., 0]).all(), iso
        assert (got["cell_flag"][iso] == ref["cell_flag"]).all(), iso
        assert (got["t"][iso].view(np.uint32) == ref["t"].view(np.uint32)).all(), iso
        assert (got["checked"][iso] == ref["checked"]).all() and (got["node"][iso] == ref["node"]).all(), iso
        st = ref["stats"]
        assert tuple(int(x) for x in got["sums"][iso][:6]) == (st.nodes_checked, st.hit_count, st.miss_count, st.oob_count,
                                                              st.escape_help_events, st.moves), iso
        n_active = W * H - st.oob_count
    assert got["traces"] < 256 * n_active / 3  # the sharing is real: far fewer traversals than ray x iso pairs
    if share:
        plain = emu_sweep(emu_lib, nodes, rd, pos, forw, up, W, H, 2.0, 0, 256, chunk=chunk, max_trips=6)
        assert got["traces"] == plain["traces"] and got["carried_on"] > got["traces"] // 3
        if not ahead:
            assert got["node_reads"] < 0.9 * plain["node_reads"]  # and so are the shared beginnings


def test_shared_traversals_on_noise_and_partial_ranges(oracle, emu_lib):
    from helpers import emu_sweep
    rng = np.random.default_rng(8)
    vox = random_volume(rng, (9, 12, 7), "noise")
    nodes, rd, _ = oracle.build_kdt(vox, "AlwaysCenter")
    pos, forw, up = oracle.main_camera(vox.shape, (50.0, -120.0, 30.0))
    for b, n, chunk, share in [(0, 256, 0, 0), (17, 100, 16, 0), (255, 1, 0, 2), (200, 56, 5, 0), (0, 256, 0, 2), (17, 100, 16, 1),
                               (3, 250, 40, 3)]:
        got = emu_sweep(emu_lib, nodes, rd, pos, forw, up, 24, 16, 0.15, b, n, chunk=chunk, share=share, ahead=n & 1,
                        max_trips=5 if share else 64)
        for j in range(n):
            ref = oracle.render(nodes, rd, pos, forw, up, 24, 16, 0.15, b + j)
            assert (got["gray"][j] == ref["rgb"][..., 0]).all() and (got["checked"][j] == ref["checked"]).all(), (b, j)
            assert (got["t"][j].view(np.uint32) == ref["t"].view(np.uint32)).all(), (b, j)


def _check_union(oracle, got, nodes, rd, pos, forw, up, W, H, vp, b, n, step=1):
    for j in range(0, n, step):
        ref = oracle.render(nodes, rd, pos, forw, up, W, H, vp, b + j)
        hit = (ref["cell_flag"] >> 24) == 1
        assert (got["gray"][j] == ref["rgb"][..., 0]).all(), (b, j)
        assert (got["cell_flag"][j] == ref["cell_flag"]).all() and (got["node"][j] == ref["node"]).all(), (b, j)
        assert (got["t"][j].view(np.uint32)[hit] == ref["t"].view(np.uint32)[hit]).all(), (b, j)
        st = ref["stats"]
        assert tuple(int(x) for x in got["sums"][j][:6]) == (st.nodes_checked, st.hit_count, st.miss_count, st.oob_count,
                                                             st.escape_help_events, st.moves), (b, j)


@pytest.mark.parametrize("trips", [5, 1, 64])
def test_union_walk_equals_one_render_per_iso(oracle, emu_lib, c60_vox, c60_tree, trips):
    """ONE traversal per ray for all 256 iso values (the front goes wherever an iso value that is not done holds;
    excluded iso values wait where the front leaves the excluding node's box): pixels, hit cells, t bits of the hits, hit
    leaves and every per-iso counter equal the oracle's separate render of each iso value; the few iso values whose own
    MOVE parts from the front's are found and walked on their own."""
    from helpers import emu_union, emu_sweep
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    W = H = 48
    got = emu_union(emu_lib, nodes, rd, pos, forw, up, W, H, 2.0, 0, 256, max_trips=trips)
    _check_union(oracle, got, nodes, rd, pos, forw, up, W, H, 2.0, 0, 256)
    shared = emu_sweep(emu_lib, nodes, rd, pos, forw, up, W, H, 2.0, 0, 256, max_trips=5, share=2)
    assert got["node_reads"] * 4 < shared["node_reads"]  # far fewer node reads than walks shared between iso values
    assert got["left_isos"] * 50 < 256 * W * H           # and leaving the union walk is rare


def test_union_walk_on_noise_partial_ranges_and_close_cameras(oracle, emu_lib):
    from helpers import emu_union
    rng = np.random.default_rng(11)
    for shape, kind, heur in [((9, 12, 7), "noise", "AlwaysCenter"), ((16, 16, 16), "smooth", None), ((12, 9, 14), "sparse", "AlwaysCenter")]:
        vox = random_volume(rng, shape, kind)
        nodes, rd, _ = oracle.build_kdt(vox, heur) if heur else oracle.build_kdt(vox)
        for cam, vp in [((50.0, -120.0, 30.0), 0.15), ((-30.0, -300.0, -30.0), 0.2), ((4.0, 5.0, -60.0), 0.5)]:
            pos, forw, up = oracle.main_camera(vox.shape, cam)
            for b, n in [(0, 256), (17, 100), (255, 1), (200, 56)]:
                got = emu_union(emu_lib, nodes, rd, pos, forw, up, 24, 16, vp, b, n, max_trips=3)
                _check_union(oracle, got, nodes, rd, pos, forw, up, 24, 16, vp, b, n, step=1 if n <= 100 else 3)


def test_union_walk_suspect_is_decided_by_the_largest_box_left(oracle, emu_lib, c60_vox, c60_tree, skull_vox, skull_tree):
    """A MOVE is suspect when a box the point has left does not share the front's exit plane.  The lane code asks the largest
    box left only (the exit plane of a box lies between its child's and its parent's along every axis); the emulator works the
    same question out level by level for every MOVE and counts the differences: none, on walks that do have suspect MOVEs."""
    from helpers import emu_union
    rng = np.random.default_rng(5)
    cases = [(c60_vox, c60_tree, 64, 48, 0, 256, None), (skull_vox, skull_tree, 96, 80, 0, 256, None),
             (None, None, 40, 32, 0, 256, (20, 17, 23)), (None, None, 40, 32, 90, 60, (16, 16, 16))]
    cases.append((None, None, 48, 40, 0, 256, (24, 24, 24)))
    n_suspect = n_states = 0
    for vox, tree, W, H, b, n, shape in cases:
        if shape is not None:
            vox = random_volume(rng, shape, "smooth" if shape == (24, 24, 24) else "noise")
            tree = oracle.build_kdt(vox, "AlwaysCenter")
        nodes, rd, _ = tree
        for cam in (None, (30.0, -200.0, 17.0), (4.0, 5.0, -60.0)):
            pos, forw, up = oracle.main_camera(vox.shape) if cam is None else oracle.main_camera(vox.shape, cam)
            got = emu_union(emu_lib, nodes, rd, pos, forw, up, W, H, 2.0 if cam is None else 0.4, b, n, max_trips=3)
            assert got["suspect_differs"] == 0, got
            n_suspect += got["suspect"]; n_states += got["states"]
    assert n_suspect >= 10 and n_states > 100 * n_suspect, (n_suspect, n_states)

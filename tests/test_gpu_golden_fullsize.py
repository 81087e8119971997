"""The default kernels at the sizes they are benchmarked on, against committed oracle fixtures (tests/golden/fullsize_oracle.json,
written by tests/golden/make_fullsize_golden.py from the CPU oracle): every frame of the Skull and 256^3 stand-in sweeps at
1024 x 1024 (the loop of src/main.zig:111-138) and the Skull frame at 4096 x 4096, by the SHA-256 of the RGB24 frame left in device
memory and all seven counters; and two span-heuristic trees of Skull (src/SlimKDT.zig:222-322) built on the GPU, by the SHA-256
of the node array and PartitionBuildData."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import volumerenderer_amd as vr
from volumerenderer_amd import capi
from test_gpu_parity import skull_renderer, standin_tree, standin_renderer  # noqa: F401

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_oracle.json")) as _f:
    GOLD = json.load(_f)


def _render_default(rnd, shape, gold, oracle):
    """The call of the fixture through vr_render_device with the default kernel selection: (frames on the device, counters)."""
    import torch
    L = capi.lib()
    W, H = gold["width"], gold["height"]
    isos = [f["iso"] for f in gold["frames"]]
    assert isos == list(range(isos[0], isos[0] + len(isos)))
    pos, forw, up = oracle.main_camera(shape)
    v = vr.raytrace._view(pos, forw, up, W, H, gold["viewport_width"])
    rnd.set_kernel(capi.KERNEL_DEFAULT)
    d_rgb = torch.full((len(isos), H, W, 3), 3, dtype=torch.uint8, device="cuda")
    d_stats = torch.zeros((len(isos), 7), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()  # the fills run on torch's stream, the render on the context's own
    capi.check(L.vr_render_device(rnd.handle, C.byref(v), isos[0], len(isos), 0, H, d_rgb.data_ptr(), d_stats.data_ptr(), None, None),
               "vr_render_device")
    capi.check(L.vr_sync(rnd.handle), "vr_sync")
    return d_rgb, d_stats.cpu().numpy()


def _assert_golden(d_rgb, stats, gold, what):
    assert gold["stat_fields"] == ["nodes_checked", "hit_count", "miss_count", "oob_count", "escape_help_events", "moves", "errors"]
    want = np.array([f["stats"] for f in gold["frames"]], dtype=np.int64)
    assert (stats == want).all(), "%s: counters differ from the oracle's at (frame, counter) %s" % (what, np.argwhere(stats != want)[:8].tolist())
    host = d_rgb.cpu().numpy()
    bad = [f["iso"] for j, f in enumerate(gold["frames"]) if hashlib.sha256(host[j].tobytes()).hexdigest() != f["sha256"]]
    assert not bad, "%s: frames differ from the oracle's at iso %s" % (what, bad[:16])


def test_default_kernel_skull_sweep_1024_equals_oracle_fixture(oracle, skull_vox, skull_renderer):
    """All 256 frames of the headline workload, on a first call and on the repeated call (tiles started longest first)."""
    gold = GOLD["skull_1024"]
    assert hashlib.sha256(skull_vox.tobytes()).hexdigest() == gold["volume_sha256"]
    for call in ("first call", "repeated call", "third call"):
        d_rgb, stats = _render_default(skull_renderer, skull_vox.shape, gold, oracle)
        _assert_golden(d_rgb, stats, gold, "Skull 1024^2 sweep, " + call)
    si = capi.SweepInfo()
    capi.check(capi.lib().vr_sweep_info_read(skull_renderer.handle, C.byref(si)), "vr_sweep_info_read")
    assert si.walks == si.items and si.tiles_longest_first == 1 and si.rendered_again == 0  # it was the union kernel, in its fast order


def test_default_kernel_skull_frame_4096_equals_oracle_fixture(oracle, skull_vox, skull_renderer):
    gold = GOLD["skull_4096"]
    d_rgb, stats = _render_default(skull_renderer, skull_vox.shape, gold, oracle)
    _assert_golden(d_rgb, stats, gold, "Skull 4096^2 iso 80")


def test_default_kernel_standin_sweep_1024_equals_oracle_fixture(oracle, standin_tree, standin_renderer):
    vox, _ = standin_tree
    gold = GOLD["standin_1024"]
    if hashlib.sha256(vox.tobytes()).hexdigest() != gold["volume_sha256"]:
        pytest.skip("this machine's numpy generates another stand-in volume than the one the fixture was rendered from")
    for call in ("first call", "repeated call", "third call"): # from the repeated call on the first tiles of the order have a finer cut
        d_rgb, stats = _render_default(standin_renderer, vox.shape, gold, oracle)
        _assert_golden(d_rgb, stats, gold, "256^3 stand-in 1024^2 sweep, " + call)
    si = capi.SweepInfo()
    capi.check(capi.lib().vr_sweep_info_read(standin_renderer.handle, C.byref(si)), "vr_sweep_info_read")
    assert si.walks == si.items and si.tiles_longest_first == 1 and si.rendered_again == 0


def test_gpu_built_span_trees_of_skull_equal_oracle_fixture(skull_vox):
    """The oracle needs minutes per tree here, so its result comes from the fixture."""
    with vr.Renderer(0) as r:
        r.upload_volume(vr.Data(skull_vox))
        for t in GOLD["skull_trees"]:
            bs = r.build_kdt(t["heuristic"])
            nodes, _ = r.download_kdt()
            used = int(np.flatnonzero(nodes)[-1]) + 1
            assert used == t["used_slots"], t["heuristic"]
            assert hashlib.sha256(nodes[:used].tobytes()).hexdigest() == t["sha256"], t["heuristic"]
            assert (bs.layer_count, bs.number_of_nodes, bs.non_center_partition_count) == (
                t["layer_count"], t["number_of_nodes"], t["non_center_partition_count"]), t["heuristic"]

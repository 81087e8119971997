"""One ray, step by step: the reference's per-ray debug print (traceRay with dbp = true, src/raytrace.zig:133-198) as
records.  The oracle's records agree with its own render (CPU); the GPU's records are the oracle's, byte for byte,
and its text output has the reference's layout."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import volumerenderer_amd as vr
from volumerenderer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 160, 120


def _pixels(ref):
    """a hit, a miss, an OOB pixel and the ray with the most node checks"""
    flag = ref["cell_flag"] >> 24
    picks = []
    for f in (1, 0, 2):
        ys, xs = np.nonzero(flag == f)
        picks.append((int(xs[len(xs) // 2]), int(ys[len(ys) // 2])))
    y, x = np.unravel_index(int(ref["checked"].argmax()), ref["checked"].shape)
    picks.append((int(x), int(y)))
    return picks


def test_oracle_ray_records_agree_with_its_render(oracle, skull_vox, skull_tree):
    nodes, rd, _ = skull_tree
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    ref = oracle.render(nodes, rd, pos, forw, up, W, H, 2.0, 80)
    for x, y in _pixels(ref):
        rep, steps = oracle.trace_ray(nodes, rd, pos, forw, up, W, H, 2.0, 80, x, y)
        assert rep.flag == ref["cell_flag"][y, x] >> 24 and rep.n_steps == ref["checked"][y, x] == len(steps)
        assert np.float32(rep.t_final).view(np.uint32) == ref["t"][y, x].view(np.uint32)
        if rep.flag == 1:  # the last record is the hit leaf, reached at the final t
            last = oracle.RayStep.from_buffer_copy(steps[-1].tobytes())
            assert last.action == 1 and last.size == 1 and last.node == ref["node"][y, x] and last.min <= 80 <= last.max
            assert np.float32(last.t).view(np.uint32) == ref["t"][y, x].view(np.uint32)
        if rep.flag == 2:
            assert rep.n_steps == 0


@pytest.mark.gpu
def test_gpu_ray_records_equal_the_oracle(oracle, skull_vox, skull_tree, tmp_path):
    nodes, rd, _ = skull_tree
    pos, forw, up = oracle.main_camera(skull_vox.shape)
    ref = oracle.render(nodes, rd, pos, forw, up, W, H, 2.0, 80)
    L = capi.lib()
    v = vr.raytrace._view(pos, forw, up, W, H, 2.0)
    with vr.Renderer(0) as r:
        r.upload_kdt(vr.SlimKDT(nodes, rd))
        for x, y in _pixels(ref):
            want_rep, want = oracle.trace_ray(nodes, rd, pos, forw, up, W, H, 2.0, 80, x, y)
            cap = 1 << 12
            steps = (capi.RayStep * cap)()
            rep = capi.RayReport()
            capi.check(L.vr_trace_ray(r.handle, C.byref(v), 80, x, y, steps, cap, C.byref(rep)), "vr_trace_ray")
            assert bytes(rep) == bytes(want_rep), (x, y)
            got = np.frombuffer(steps, dtype=np.uint8).reshape(cap, C.sizeof(capi.RayStep))[: rep.n_stored]
            assert (got == want).all(), (x, y)
            # a short buffer keeps the first records and still counts all iterations
            few = (capi.RayStep * 3)()
            rep2 = capi.RayReport()
            capi.check(L.vr_trace_ray(r.handle, C.byref(v), 80, x, y, few, 3, C.byref(rep2)), "vr_trace_ray")
            assert rep2.n_steps == rep.n_steps and rep2.n_stored == min(3, rep.n_steps)
        path = str(tmp_path / "ray.txt")
        capi.check(L.vr_print_ray_trace(C.byref(rep), steps, path.encode()), "vr_print_ray_trace")
        text = open(path).read()
        assert "### NEW RAY ###" in text and text.count("node=") == rep.n_stored and "skip to (" in text
        assert text.count("MOVE!") + text.count("ZOOM!") + text.count("HIT!") == rep.n_stored
        assert L.vr_trace_ray(r.handle, C.byref(v), 80, W, 0, None, 0, C.byref(rep)) == capi.VR_EARG


@pytest.mark.gpu
def test_environment_variable_prints_the_ray_of_a_render():
    code = ("import os, sys; sys.path.insert(0, %r); import volumerenderer_amd as vr\n"
            "d = vr.Data.loadUCD(os.path.join(%r, 'tests', 'data', 'C60.vol')); k = vr.SlimKDT.newHeursiticPartition(d)\n"
            "pos, forw, up = vr.main_camera(d.resulution); img = vr.Image(64, 64)\n"
            "with vr.Renderer(0) as r:\n    r.upload_kdt(k); print(r.renderImage(pos, forw, up, img, 2.0, 100)['nodes_checked'])\n" % (ROOT, ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, VR_TRACE_RAY="32,32"))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "### NEW RAY ###" in out.stderr and "iso=[" in out.stderr and " v 100" in out.stderr

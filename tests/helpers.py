import ctypes as C

import numpy as np


def prepare_nodes(nodes, rd):
    """vr_host_prepare_kdt of the product library (host-only): validated node words, 1-based heap."""
    from volumerenderer_amd import capi
    L = capi.lib()
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    rd3 = (C.c_uint8 * 3)(*[int(x) for x in rd])
    depth = C.c_uint32()
    count = C.c_uint64()
    out = np.zeros(L.vr_prepared_words(len(nodes)), dtype=np.uint32)
    rc = L.vr_host_prepare_kdt(nodes.ctypes.data, len(nodes), rd3, out.ctypes.data, len(out), C.byref(depth),
                               C.byref(count))
    return rc, out, depth.value, count.value


def emu_render(L, variant, nodes, rd, pos, f, u, w, h, vp, iso):
    """Runs the kernels' lane code on the CPU (tests/emu/vr_emu.cpp)."""
    n_slots = len(nodes)
    if variant != 1:
        rc, nodes, _, _ = prepare_nodes(nodes, rd)
        assert rc == 0
    rgb = np.zeros((h, w, 3), np.uint8)
    cf = np.zeros((h, w), np.uint32)
    tt = np.zeros((h, w), np.float32)
    ck = np.zeros((h, w), np.uint32)
    nd = np.zeros((h, w), np.uint32)
    sums = np.zeros(7, np.uint64)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    L.vr_emu_render(int(variant), C.c_void_p(nodes.ctypes.data), C.c_uint64(n_slots),
                    (C.c_uint8 * 3)(*[int(x) for x in rd]), f3(pos), f3(f), f3(u), C.c_uint32(w), C.c_uint32(h),
                    C.c_float(vp), C.c_uint32(iso), C.c_void_p(rgb.ctypes.data), C.c_void_p(cf.ctypes.data),
                    C.c_void_p(tt.ctypes.data), C.c_void_p(ck.ctypes.data), C.c_void_p(nd.ctypes.data),
                    C.c_void_p(sums.ctypes.data))
    return {"rgb": rgb, "cell_flag": cf, "t": tt, "checked": ck, "node": nd, "sums": sums}


def emu_wide(L, nodes, rd, pos, f, u, w, h, vp, iso, pop_trips=0, ahead_trips=0):
    """The resume traversal of wide volumes (vr_trace_wide.h) on the CPU, on wide node words; pop_trips bounds every
    pop phase as the persistent kernel does (0 = unbounded)."""
    rgb = np.zeros((h, w, 3), np.uint8)
    cf = np.zeros((h, w), np.uint32)
    tt = np.zeros((h, w), np.float32)
    ck = np.zeros((h, w), np.uint32)
    nd = np.zeros((h, w), np.uint32)
    sums = np.zeros(7, np.uint64)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    L.vr_emu_wide(C.c_void_p(nodes.ctypes.data), C.c_uint64(len(nodes)), (C.c_uint16 * 3)(*[int(x) for x in rd]),
                  f3(pos), f3(f), f3(u), C.c_uint32(w), C.c_uint32(h), C.c_float(vp), C.c_uint32(iso),
                  C.c_uint32(pop_trips), C.c_uint32(ahead_trips), C.c_void_p(rgb.ctypes.data), C.c_void_p(cf.ctypes.data), C.c_void_p(tt.ctypes.data),
                  C.c_void_p(ck.ctypes.data), C.c_void_p(nd.ctypes.data), C.c_void_p(sums.ctypes.data))
    return {"rgb": rgb, "cell_flag": cf, "t": tt, "checked": ck, "node": nd, "sums": sums}


def emu_sweep(L, nodes, rd, pos, f, u, w, h, vp, iso_begin, iso_count, chunk=0, max_trips=64, share=0, ahead=0):
    """The iso sweep with traversals shared between iso values, on the CPU (tests/emu/vr_emu.cpp); `share` = lane
    states kept for later walks of the same ray to carry on from, `ahead` = descents that read both children ahead."""
    n_slots = len(nodes)
    rc, prep, _, _ = prepare_nodes(nodes, rd)
    assert rc == 0
    gray = np.zeros((iso_count, h, w), np.uint8)
    cf = np.zeros((iso_count, h, w), np.uint32)
    tt = np.zeros((iso_count, h, w), np.float32)
    ck = np.zeros((iso_count, h, w), np.uint32)
    nd = np.zeros((iso_count, h, w), np.uint32)
    sums = np.zeros((iso_count, 7), np.uint64)
    counts = np.zeros(5, np.uint64)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
    prep = np.ascontiguousarray(prep, dtype=np.uint32)
    L.vr_emu_sweep(C.c_void_p(prep.ctypes.data), C.c_uint64(n_slots), (C.c_uint8 * 3)(*[int(x) for x in rd]), f3(pos),
                   f3(f), f3(u), C.c_uint32(w), C.c_uint32(h), C.c_float(vp), C.c_uint32(iso_begin),
                   C.c_uint32(iso_count), C.c_uint32(chunk), C.c_uint32(max_trips), C.c_uint32(share), C.c_uint32(ahead), C.c_void_p(gray.ctypes.data),
                   C.c_void_p(cf.ctypes.data), C.c_void_p(tt.ctypes.data), C.c_void_p(ck.ctypes.data),
                   C.c_void_p(nd.ctypes.data), C.c_void_p(sums.ctypes.data), C.c_void_p(counts.ctypes.data))
    return {"gray": gray, "cell_flag": cf, "t": tt, "checked": ck, "node": nd, "sums": sums, "traces": int(counts[0]),
            "node_reads": int(counts[1]), "carried_on": int(counts[2]), "returns_walked": int(counts[3]),
            "returns_distinct": int(counts[4])}


def emu_union(L, nodes, rd, pos, f, u, w, h, vp, iso_begin, iso_count, max_trips=5):
    """The sweep as ONE traversal per ray for all iso values (vr_trace_union.h), on the CPU (tests/emu/vr_emu.cpp)."""
    n_slots = len(nodes)
    rc, prep, _, _ = prepare_nodes(nodes, rd)
    assert rc == 0
    gray = np.zeros((iso_count, h, w), np.uint8)
    cf = np.zeros((iso_count, h, w), np.uint32)
    tt = np.zeros((iso_count, h, w), np.float32)
    nd = np.zeros((iso_count, h, w), np.uint32)
    sums = np.zeros((iso_count, 7), np.uint64)
    counts = np.zeros(6, np.uint64)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
    prep = np.ascontiguousarray(prep, dtype=np.uint32)
    L.vr_emu_union(C.c_void_p(prep.ctypes.data), C.c_uint64(n_slots), (C.c_uint8 * 3)(*[int(x) for x in rd]), f3(pos),
                   f3(f), f3(u), C.c_uint32(w), C.c_uint32(h), C.c_float(vp), C.c_uint32(iso_begin), C.c_uint32(iso_count),
                   C.c_uint32(max_trips), C.c_void_p(gray.ctypes.data), C.c_void_p(cf.ctypes.data), C.c_void_p(tt.ctypes.data),
                   C.c_void_p(nd.ctypes.data), C.c_void_p(sums.ctypes.data), C.c_void_p(counts.ctypes.data))
    return {"gray": gray, "cell_flag": cf, "t": tt, "node": nd, "sums": sums, "node_reads": int(counts[0]),
            "states": int(counts[1]), "suspect": int(counts[2]), "left_runs": int(counts[3]), "left_isos": int(counts[4]), "suspect_differs": int(counts[5])}


def emu_union_wide(L, nodes, rd, pos, f, u, w, h, vp, iso_begin, iso_count, max_trips=5):
    """The union walk on a wide tree (vr_trace_union_wide.h), on the CPU."""
    gray = np.zeros((iso_count, h, w), np.uint8)
    cf = np.zeros((iso_count, h, w), np.uint32)
    tt = np.zeros((iso_count, h, w), np.float32)
    nd = np.zeros((iso_count, h, w), np.uint32)
    sums = np.zeros((iso_count, 7), np.uint64)
    counts = np.zeros(5, np.uint64)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    L.vr_emu_union_wide(C.c_void_p(nodes.ctypes.data), C.c_uint64(len(nodes)), (C.c_uint16 * 3)(*[int(x) for x in rd]), f3(pos),
                        f3(f), f3(u), C.c_uint32(w), C.c_uint32(h), C.c_float(vp), C.c_uint32(iso_begin), C.c_uint32(iso_count),
                        C.c_uint32(max_trips), C.c_void_p(gray.ctypes.data), C.c_void_p(cf.ctypes.data), C.c_void_p(tt.ctypes.data),
                        C.c_void_p(nd.ctypes.data), C.c_void_p(sums.ctypes.data), C.c_void_p(counts.ctypes.data))
    return {"gray": gray, "cell_flag": cf, "t": tt, "node": nd, "sums": sums, "node_reads": int(counts[0]),
            "states": int(counts[1]), "suspect": int(counts[2]), "left_runs": int(counts[3]), "left_isos": int(counts[4])}


def assert_same_render(got, ref, what=""):
    """Bit-exact comparison of every per-pixel output against the oracle."""
    assert (got["rgb"] == ref["rgb"]).all(), "%s: pixels differ (%d)" % (what, int((got["rgb"] != ref["rgb"]).sum()))
    assert (got["cell_flag"] == ref["cell_flag"]).all(), "%s: hit cells / flags differ" % what
    assert (got["t"].view(np.uint32) == ref["t"].view(np.uint32)).all(), "%s: t bit patterns differ" % what
    assert (got["checked"] == ref["checked"]).all(), "%s: checked_nodes differ" % what
    assert (got["node"] == ref["node"]).all(), "%s: hit leaf indices differ" % what


def random_volume(rng, shape, kind):
    if kind == "noise":
        return rng.integers(0, 256, size=shape, dtype=np.uint8)
    if kind == "smooth":
        g = np.stack(np.meshgrid(*[np.linspace(-1, 1, s) for s in shape], indexing="ij"))
        r = np.sqrt((g ** 2).sum(0))
        return np.clip(255 * np.exp(-3 * r * r) + rng.normal(0, 6, shape), 0, 255).astype(np.uint8)
    if kind == "sparse":
        v = np.zeros(shape, np.uint8)
        m = rng.random(shape) < 0.03
        v[m] = rng.integers(1, 256, size=int(m.sum()), dtype=np.uint8)
        return v
    raise ValueError(kind)


def fullsize_golden_stats(name):
    """The oracle's seven counters per frame of one of the full-size fixtures (tests/golden/fullsize_oracle.json, written by
    tests/golden/make_fullsize_golden.py): int64 [frames, 7] in the order of vr_render_stats."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_oracle.json")) as f:
        gold = json.load(f)[name]
    return np.array([fr["stats"] for fr in gold["frames"]], dtype=np.int64)

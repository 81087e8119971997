"""The CPU oracle against the reference's own committed renders (the only results the reference pins)."""
import hashlib
import json
import os

import numpy as np

from conftest import GOLDEN


def test_oracle_reproduces_all_256_reference_renders(oracle, c60_vox, c60_tree, golden_gray):
    nodes, rd, st = c60_tree
    assert st.layer_count == 18 and st.number_of_nodes == 500093 and st.non_center_partition_count == 0
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    meta = json.load(open(os.path.join(GOLDEN, "c60_renders_512.json")))
    for iso in range(256):
        r = oracle.render(nodes, rd, pos, forw, up, 512, 512, 2.0, iso, taps=False)
        rgb = r["rgb"]
        assert (rgb[..., 0] == golden_gray[iso]).all(), "iso %d: %d pixels differ" % (
            iso, int((rgb[..., 0] != golden_gray[iso]).sum()))
        assert (rgb[..., 0] == rgb[..., 1]).all() and (rgb[..., 1] == rgb[..., 2]).all()
        assert hashlib.sha256(rgb.tobytes()).hexdigest() == meta["isos"][iso]["sha256_rgb24"]
        assert r["stats"].oob_count == 216422  # the entry test does not depend on the iso value
        assert r["stats"].escape_help_events == 0


def test_oracle_known_counts(oracle, c60_vox, c60_tree):
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    hits = {0: 38252, 100: 28226, 255: 13262}
    for iso, n in hits.items():
        r = oracle.render(nodes, rd, pos, forw, up, 512, 512, 2.0, iso, taps=False)
        assert r["stats"].hit_count == n
        assert r["stats"].hit_count + r["stats"].miss_count + r["stats"].oob_count == 512 * 512


def test_oracle_rows_equal_full_image(oracle, c60_vox, c60_tree):
    nodes, rd, _ = c60_tree
    pos, forw, up = oracle.main_camera(c60_vox.shape)
    full = oracle.render(nodes, rd, pos, forw, up, 96, 80, 2.0, 90)
    out = np.zeros((80, 96, 3), np.uint8)
    total = 0
    for r0, r1 in [(0, 31), (31, 32), (32, 80)]:
        part = oracle.render(nodes, rd, pos, forw, up, 96, 80, 2.0, 90, rows=(r0, r1), out=out)
        total += part["stats"].nodes_checked
    assert (out == full["rgb"]).all()
    assert total == full["stats"].nodes_checked

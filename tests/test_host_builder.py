"""Host-side producers of libvr_hip.so (loader, SlimKDT builder, camera) against the oracle.  No GPU."""
import os
import struct

import numpy as np
import pytest

import volumerenderer_amd as vr
from volumerenderer_amd import capi
from conftest import DATA
from helpers import random_volume

LIST_OF_MAIN = ["AlwaysCenter", "FlexibleFirst5", "FlexibleAfter15", "FlexibleAfter20", "FlexibleCenterWeighted30",
                "FlexibleCenterWeighted50", "FlexibleCenterWeighted75"]  # src/main.zig:60-75


def _same_tree(k, nodes_o, st_o):
    n = len(k.nodes)
    ref = nodes_o[:n] if len(nodes_o) >= n else np.concatenate([nodes_o, np.zeros(n - len(nodes_o), np.uint32)])
    assert (k.nodes == ref).all()
    assert not nodes_o[n:].any()
    assert k.build.layer_count == st_o.layer_count
    assert k.build.number_of_nodes == st_o.number_of_nodes
    assert k.build.non_center_partition_count == st_o.non_center_partition_count


def test_loader_matches_oracle_and_layout(oracle, tmp_path):
    d = vr.Data.loadUCD(os.path.join(DATA, "C60.vol"))
    assert d.resulution == (64, 64, 64)
    assert (d.data == oracle.load_ucd(os.path.join(DATA, "C60.vol"))).all()
    # x slowest, z fastest; big-endian header; trailing header words ignored
    vox = np.arange(2 * 3 * 5, dtype=np.uint8).reshape(2, 3, 5)
    p = tmp_path / "t.vol"
    p.write_bytes(struct.pack(">7I", 2, 3, 5, 7, 1, 2, 3) + vox.tobytes())
    d = vr.Data.loadUCD(str(p))
    assert d.resulution == (2, 3, 5) and d.get(1, 2, 4) == 29 and d.get(1, 0, 0) == 15
    p.write_bytes(struct.pack(">7I", 2, 3, 5, 0, 0, 0, 0) + vox.tobytes()[:-1])
    with pytest.raises(capi.VrError) as e:
        vr.Data.loadUCD(str(p))
    assert e.value.code == capi.VR_EFORMAT
    with pytest.raises(capi.VrError) as e:
        vr.Data.loadUCD(str(tmp_path / "missing.vol"))
    assert e.value.code == capi.VR_EIO


def test_skull_gz_loads(skull_vox):
    d = vr.Data.loadUCD(os.path.join(DATA, "Skull.vol.gz"))
    assert d.resulution == (68, 256, 256)
    assert (d.data == skull_vox).all()


@pytest.mark.parametrize("heuristic", LIST_OF_MAIN)
def test_c60_tree_matches_oracle(oracle, c60_vox, heuristic):
    k = vr.SlimKDT.newHeursiticPartition(vr.Data(c60_vox), heuristic)
    nodes_o, rd_o, st_o = oracle.build_kdt(c60_vox, heuristic)
    assert (k.root_dim == rd_o).all()
    _same_tree(k, nodes_o, st_o)


def test_skull_center_tree_matches_oracle(skull_vox, skull_tree):
    nodes_o, rd_o, st_o = skull_tree
    k = vr.SlimKDT.newHeursiticPartition(vr.Data(skull_vox), "AlwaysCenter")
    assert st_o.layer_count == 23 and st_o.number_of_nodes == 8713349
    _same_tree(k, nodes_o, st_o)


@pytest.mark.parametrize("shape", [(2, 2, 2), (2, 3, 9), (17, 5, 12), (9, 31, 8), (24, 24, 24)])
@pytest.mark.parametrize("kind", ["noise", "smooth", "sparse"])
def test_small_volumes_every_heuristic(oracle, shape, kind):
    rng = np.random.default_rng(sum(shape) * 31 + len(kind))
    vox = random_volume(rng, shape, kind)
    for h in capi.HEURISTICS:
        try:
            nodes_o, rd_o, st_o = oracle.build_kdt(vox, h, max_slots=1 << 24)
        except RuntimeError:
            with pytest.raises(capi.VrError) as e:
                vr.SlimKDT.newHeursiticPartition(vr.Data(vox), h, max_slots=1 << 24)
            assert e.value.code == capi.VR_ETOODEEP
            continue
        k = vr.SlimKDT.newHeursiticPartition(vr.Data(vox), h, max_slots=1 << 24)
        _same_tree(k, nodes_o, st_o)


def test_too_deep_tree_is_reported(c60_vox):
    # the unbalanced span-only tree outgrows the heap array: the reference's "FAILD TO BUILD KDT"
    with pytest.raises(capi.VrError) as e:
        vr.SlimKDT.newHeursiticPartition(vr.Data(c60_vox), "MinimizeSpan", max_slots=1 << 26)
    assert e.value.code == capi.VR_ETOODEEP


def test_builder_rejects_bad_arguments():
    with pytest.raises(capi.VrError):
        vr.SlimKDT.newHeursiticPartition(vr.Data(np.zeros((1, 4, 4), np.uint8)))
    with pytest.raises(capi.VrError):
        vr.SlimKDT.newHeursiticPartition(vr.Data(np.zeros((4, 4, 4), np.uint8)), 99)


def test_main_camera_matches_oracle(oracle):
    for res in [(64, 64, 64), (68, 256, 256), (3, 200, 17)]:
        for pos in [(-30.0, -300.0, -30.0), (400.0, 10.0, -7.5)]:
            a = vr.main_camera(res, pos)
            b = oracle.main_camera(res, pos)
            for x, y in zip(a, b):
                assert (x.view(np.uint32) == y.view(np.uint32)).all()


def test_prepare_kdt_flattens_and_validates(oracle, c60_tree):
    from helpers import prepare_nodes
    nodes, rd, st = c60_tree
    rc, prep, depth, count = prepare_nodes(nodes, rd)
    assert rc == 0 and depth == st.layer_count and count == st.number_of_nodes
    assert len(prep) == len(nodes) + 1 and prep[0] == 0                  # 1-based heap: position 0 is unused
    prep = prep[1:]
    assert ((prep & 0x00FFFFFF) == (nodes & 0x00FFFFFF)).all()          # min, max, partition untouched
    used = prep != 0
    leaf = (prep >> 31) == 1
    assert int(leaf.sum()) == 63 ** 3                                    # one leaf per cell
    inner = used & ~leaf
    assert (((prep[inner] >> 24) & 31) == 10 * ((nodes[inner] >> 24) & 3)).all()
    # undefined upper bits of the Zig packed struct are ignored
    noisy = nodes | (np.uint32(0xFC) << 24)
    rc2, prep2, _, _ = prepare_nodes(noisy, rd)
    assert rc2 == 0 and (prep2[1:] == prep).all()
    # malformed trees are rejected: partition on the box edge, bad axis, truncated array
    bad = nodes.copy(); bad[0] = (bad[0] & ~np.uint32(0x00FF0000)) | (np.uint32(63) << 16)
    assert prepare_nodes(bad, rd)[0] == capi.VR_ETREE
    bad = nodes.copy(); bad[0] = bad[0] & ~np.uint32(0x00FF0000)
    assert prepare_nodes(bad, rd)[0] == capi.VR_ETREE
    bad = nodes.copy(); bad[5] |= np.uint32(3 << 24)
    assert prepare_nodes(bad, rd)[0] == capi.VR_ETREE
    assert prepare_nodes(nodes[: len(nodes) // 4], rd)[0] == capi.VR_ETREE

"""Cascade loading: known answers from the reference's XML text, .vjc round trips,
and the product's C++ loader against the oracle's independent readers."""
import ctypes as C
import os

import numpy as np
import pytest

from clfacedetection_amd import Cascade, VjError, load_library
from clfacedetection_amd.api import DATA_DIR
from oracle.oracle import load_vjc, parse_xml

REF = "/root/reference/CLFaceDetection"
NAMES = ["frontalface_default", "frontalface_alt", "frontalface_alt2", "frontalface_alt_tree", "eye"]
TILTED = ["fullbody", "eye_tree_eyeglasses"]   # shipped for the OpenCV profile's tilted-feature tests
# win, stages, trees, nodes, maxT, maxN, tilted, 3-rect
TABLE = {
    "frontalface_default": (24, 25, 2913, 2913, 211, 1, 0, 557),
    "frontalface_alt": (20, 22, 2135, 2135, 213, 1, 0, 360),
    "frontalface_alt2": (20, 20, 1047, 2094, 109, 2, 0, 347),
    "frontalface_alt_tree": (20, 47, 8468, 8468, 406, 1, 0, 1545),
    "eye": (20, 24, 1066, 1066, 93, 1, 0, 167),
}


@pytest.mark.parametrize("name", NAMES)
def test_structure_matches_survey_table(name):
    i = Cascade.load(name).info
    assert (i.win_w, i.n_stages, i.n_trees, i.n_nodes, i.max_trees_per_stage, i.max_nodes_per_tree, i.n_tilted,
            i.n_three_rect) == TABLE[name]
    assert i.win_h == i.win_w
    assert i.is_stump_based == (name not in ("frontalface_alt2",))
    assert i.is_stage_tree == (name == "frontalface_alt_tree")


def test_frontalface_alt_known_answers():
    """Constants read straight from haarcascade_frontalface_alt.xml (stage 0, lines 50-91)."""
    c = Cascade.load("frontalface_alt")
    st, nd, al = c.stages, c.nodes, c.alpha
    assert st["n_trees"].tolist() == [3, 16, 21, 39, 33, 44, 50, 51, 56, 71, 80, 103, 111, 102, 135, 137, 140, 160,
                                      177, 182, 211, 213]
    assert st["threshold"][0] == np.float32(0.8226894140243530)
    assert (st["parent"][0], st["next"][0], st["child"][0]) == (-1, -1, 1)
    r = nd["rect"][0]
    assert [tuple(int(r[k][f]) for f in "xywh") for k in range(2)] == [(3, 7, 14, 4), (3, 9, 14, 2)]
    assert r["weight"].tolist() == [-1.0, 2.0, 0.0]
    assert nd["threshold"][0] == np.float32(4.0141958743333817e-003)
    assert (al[0], al[1]) == (np.float32(0.0337941907346249), np.float32(0.8378106951713562))
    assert (nd["left"][0], nd["right"][0]) == (0, -1)
    # tree 1 of stage 0 has a weight-3 second rectangle
    assert nd["rect"][1]["weight"].tolist() == [-1.0, 3.0, 0.0]


def test_alt_tree_stage_links():
    """stage 4 has children 5 and 6 (5.next = 6); chains 5->7->..->39 and 6->8->..->46."""
    st = Cascade.load("frontalface_alt_tree").stages
    assert st["child"][4] == 5 and st["parent"][5] == 4 and st["parent"][6] == 4 and st["next"][5] == 6
    assert st["child"][5] == 7 and st["child"][6] == 8 and st["child"][39] == -1 and st["child"][46] == -1
    assert (st["next"] != -1).sum() == 1


def test_alt2_two_node_trees():
    c = Cascade.load("frontalface_alt2")
    tr, nd = c.trees, c.nodes
    assert set(tr["n_nodes"].tolist()) == {2}
    root = nd[tr["first_node"]]
    # exactly one of left/right of every root points at node 1 (haarcascade_frontalface_alt2.xml)
    assert (((root["left"] == 1) ^ (root["right"] == 1))).all()


@pytest.mark.parametrize("name", NAMES)
def test_vjc_matches_independent_reader(name):
    c = Cascade.load(name)
    a = load_vjc(os.path.join(DATA_DIR, f"haarcascade_{name}.vjc"))
    nd = c.nodes
    assert np.array_equal(nd["rect"]["weight"].reshape(-1).view(np.uint32), a.node_weight.view(np.uint32))
    assert np.array_equal(np.stack([nd["rect"][f] for f in "xywh"], -1).reshape(-1), a.node_rect)
    assert np.array_equal(c.alpha.view(np.uint32), a.alpha.view(np.uint32))
    assert np.array_equal(c.stages["child"], a.stage_child)
    assert "Intel License Agreement" in c.notice or "license" in c.notice.lower()


@pytest.mark.parametrize("name", NAMES)
def test_vjc_round_trip(tmp_path, name):
    c = Cascade.load(name)
    p = str(tmp_path / "rt.vjc")
    c.save(p)
    assert open(p, "rb").read() == open(os.path.join(DATA_DIR, f"haarcascade_{name}.vjc"), "rb").read()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (e.g. on the GPU box)")
@pytest.mark.parametrize("xml", sorted(f for f in (os.listdir(REF) if os.path.isdir(REF) else []) if f.endswith(".xml")))
def test_xml_loader_matches_oracle_parser(tmp_path, xml):
    """All 19 stock XMLs: the C++ XML reader and the oracle's ElementTree reader agree bit for bit."""
    c = Cascade.load_xml(os.path.join(REF, xml))
    p = str(tmp_path / "x.vjc")
    c.save(p)
    assert parse_xml(os.path.join(REF, xml)).same_as(load_vjc(p)) == []


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
@pytest.mark.parametrize("name", NAMES + TILTED)
def test_shipped_vjc_is_current(name):
    a = parse_xml(os.path.join(REF, f"haarcascade_{name}.xml"))
    assert a.same_as(load_vjc(os.path.join(DATA_DIR, f"haarcascade_{name}.vjc"))) == []


def test_loader_errors(tmp_path):
    with pytest.raises(VjError) as e:
        Cascade.load(str(tmp_path / "missing.vjc"))
    assert e.value.code == 2
    bad = tmp_path / "bad.vjc"
    bad.write_bytes(b"not a cascade")
    with pytest.raises(VjError) as e:
        Cascade.load(str(bad))
    assert e.value.code == 3
    good = open(os.path.join(DATA_DIR, "haarcascade_eye.vjc"), "rb").read()
    trunc = tmp_path / "trunc.vjc"
    trunc.write_bytes(good[:-100])
    with pytest.raises(VjError):
        Cascade.load(str(trunc))
    x = tmp_path / "bad.xml"
    x.write_text("<opencv_storage><c><size>20 20</size><stages><_><trees></trees></_></stages></c></opencv_storage>")
    with pytest.raises(VjError) as e:
        Cascade.load_xml(str(x))
    assert e.value.code == 3
    lib = load_library()
    assert lib.vj_cascade_load(None, None) == 1
    assert lib.vj_strerror(3) == b"malformed cascade file"

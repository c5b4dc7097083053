"""The 19 cascades the reference ships, as converted .vjc data: the product's loader and the oracle's read the same arrays, the
licence notice travels, and — where the reference checkout is present (the build container) — the XML parses to the same."""
import glob
import os

import numpy as np
import pytest

from clfacedetection_amd.api import DATA_DIR

NAMES = sorted(os.path.basename(p)[len("haarcascade_"):-4] for p in glob.glob(os.path.join(DATA_DIR, "haarcascade_*.vjc")))
REF = "/root/reference/CLFaceDetection"


def test_nineteen_cascades_are_shipped():
    assert len(NAMES) == 19 and {"frontalface_alt", "mcs_eyepair_small", "profileface", "righteye_2splits", "upperbody"} <= set(NAMES)


@pytest.mark.parametrize("name", NAMES)
def test_loaders_agree(cascades, name):
    c, a = cascades(name)
    i = c.info
    assert (i.win_w, i.win_h) == (a.win_w, a.win_h) and i.n_stages == a.n_stages and i.n_nodes == a.n_nodes
    assert i.n_tilted == int(np.count_nonzero(a.node_tilted)) and i.max_nodes_per_tree == int(a.tree_n_nodes.max())
    assert "license" in a.notice.lower() and len(a.notice) > 1000     # the licence text of the source XML (Intel's, or the contributors' agreement of the mcs_* files) is part of the file
    xml = os.path.join(REF, f"haarcascade_{name}.xml")
    if os.path.exists(xml):
        from oracle.oracle import parse_xml
        assert parse_xml(xml).same_as(a) == []


@pytest.mark.parametrize("name", ["mcs_eyepair_small", "lowerbody", "righteye_2splits"])
def test_oracle_runs_the_shapes_no_other_test_has(oracle, cascades, name):
    """Non-square windows and tilted cascades in the clod restatement (tilted rectangles as upright ones, as the reference reads
    them): the oracle finishes and both of its paths visit the same number of scales' worth of windows as its own plan says."""
    from cases import make_frame
    c, a = cascades(name)
    img = make_frame("blocks", 5, 150, 200)
    ro, st = oracle.detect(a, img)
    assert st["windows"] > 0 and st["stage_entered"][0] == st["windows"] and len(ro) <= st["windows"]
    rc, sc = oracle.detect_opencvlike(a, img)
    assert sc["windows"] > 0

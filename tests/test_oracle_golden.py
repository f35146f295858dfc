"""The oracle against the committed golden vectors (tests/golden/, made by make_golden.py)."""
import glob
import os
import re

import numpy as np
import pytest

import oracle_binding as ob
from raytrace_amd import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


def scene_of(path):
    name = os.path.basename(path)
    m = re.match(r"(.+)_(\d+)x(\d+)\.npz", name)
    key, w, h = m.group(1), int(m.group(2)), int(m.group(3))
    kw = {}
    if key == "synthetic300+30":
        world, lights = scenes.synthetic(ntri=300, ncuboid=30)
    elif key == "cylinders6+15_spp4":
        world, lights = scenes.synthetic_cylinders(ncyl=6, ncuboid=15)
        kw = dict(spp=4, seed=7)
    else:
        world, lights = scenes.REFERENCE_SCENES[key]()
    return world, lights, w, h, kw


def test_fixtures_exist():
    assert len(FILES) >= 12


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_oracle_reproduces_golden(path):
    world, lights, w, h, kw = scene_of(path)
    g = np.load(path)
    o = ob.OracleScene(world, lights).render(w, h, depth=4, want_t=True, **kw)
    assert (o["hit_id"] == g["hit_id"]).all()
    assert (o["rgb"] == g["rgb"]).all()
    assert np.array_equal(o["t"], g["t"])

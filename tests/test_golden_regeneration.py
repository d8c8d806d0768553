"""Every committed fixture under tests/golden/ must come out of the committed generator scripts.

Runs only where the reference checkout exists (the build container: the generators import the reference's own Python from
/root/reference; the GPU box does not have it): each generator script is run into a temporary directory
(TM_GOLDEN_OUT) and every array of every fixture it writes is compared, bit for bit, with the committed file.  The one slow
generator (config4: 3 minutes of the reference's dense N^2 numpy code on 6.4k atoms) only runs with TM_GOLDEN_FULL=1."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REF = os.environ.get("TM_REFERENCE", "/root/reference")

JOBS = [
    ("generate_golden.py", []),
    ("generate_golden_groups.py", []),
    ("generate_golden_next.py", ["vv", "barostat", "hrex", "edges", "box_resize", "filter"] + (["config4"] if os.environ.get("TM_GOLDEN_FULL") else [])),
]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "timemachine")), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("script,args", JOBS, ids=[j[0] for j in JOBS])
def test_committed_fixtures_come_out_of_the_committed_generators(tmp_path, script, args):
    env = dict(os.environ, TM_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, script)] + args, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    made = sorted(glob.glob(os.path.join(str(tmp_path), "*.npz")))
    assert made, "the generator wrote nothing"
    for path in made:
        name = os.path.basename(path)
        committed = os.path.join(GOLDEN, name)
        assert os.path.exists(committed), f"{name} is generated but not committed"
        new, old = np.load(path), np.load(committed)
        assert sorted(new.files) == sorted(old.files), name
        for key in new.files:
            a, b = new[key], old[key]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, key)
            assert a.tobytes() == b.tobytes(), f"{name}[{key}] differs from what {script} generates"


def test_every_committed_fixture_has_a_generator():
    """npz files nobody generates would be unpinned data"""
    generated = {"nb_small_w0", "nb_small_whalf", "nb_small_wrand", "bonded", "config2_lambda0.0", "config2_lambda0.3", "config2_lambda1.0", "integrator",
                 "hilbert", "groups", "vv", "barostat", "hrex", "edge_ortho", "edge_drift", "config1_vacuum", "config1_pbc", "edge_box_resize",
                 "filter_exclusions", "config4"}
    present = {os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))}
    assert present == generated, present ^ generated

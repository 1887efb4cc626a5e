"""Every helper under scripts/ and tests/tools/ at least parses (they run on the GPU box, where a syntax error costs a GPU call), and the
segmentation sweep's volume generator produces what it says: touching / bridged blobs the oracle splits."""
import glob
import os
import py_compile
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scripts_parse(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "scripts", "*.py")) + glob.glob(os.path.join(ROOT, "scripts", "probes", "*.py"))
                   + glob.glob(os.path.join(ROOT, "tests", "tools", "*.py")))
    assert len(files) > 10
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_fuzz_segment_volumes_exercise_the_non_convex_path():
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    try:
        import fuzz_segment as F
    finally:
        sys.path.pop(0)
    from oracle import watershed_ref as W
    rng = np.random.default_rng(0)
    kinds = set()
    for _ in range(6):
        m = F.volume(rng, 32)
        assert m.dtype == np.uint8 and m.shape == (32, 32, 32) and 0 < m.mean() < 0.5
        tr = []
        W.segment_nuclei(m.astype(np.int32), trace=tr, degenerate="solid")
        kinds |= {t[4] for t in tr}
    assert {"convex", "recurse"} <= kinds

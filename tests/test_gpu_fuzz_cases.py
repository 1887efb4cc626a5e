"""The cases tests/tools/fuzz_steps.py (randomised whole-network parity sweep) found or came close to, kept as
tests: a train step at a batch BELOW the handle's max_batch, at sizes where the plans are not monotone in the batch
(round 6: 3 grids on a 5-grid handle at 32^3 wrote past the bias-gradient partials of the two 16-channel layers).  The script
checks metrics and every gradient tensor against oracle/torch_ref.py in fp64 with the engine's decisions pinned.
Reference: /root/reference/unet/unet.py:272-355,370, vae/lattice_vae.py:160-270,296."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_steps_below_max_batch_match_the_oracle():
    cases = "vae,32,1,5,3;vae,32,4,5,3;unet,32,1,5,4;unet,32,1,6,4;vae,32,1,7,4;unet,16,4,33,17;vae,16,1,40,3"
    env = dict(os.environ, FUZZ_CASES=cases, ICSG3D_DEBUG_CANARY="1", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "fuzz_steps.py")], capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=850)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith(("  ok", "  FAIL"))]
    print("\n".join(lines))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert len(lines) == len(cases.split(";")) and all(ln.startswith("  ok") for ln in lines)
    assert "CANARY DIRTY" not in p.stderr        # the handles check their guard bytes when they are destroyed

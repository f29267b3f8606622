"""A short, seeded run of the randomised stress tools (tools/stress.py, tools/stress_grouped.py, tools/stress_split.py: the split-K that
reduces inside the launch, alternating shapes through one workspace and a second stream): run-to-run determinism of every
GEMM path and agreement between paths on random shapes, splits and weight modes.  The long runs are a tool; this keeps a slice of
them in the suite (a 4-minute run is what found the 64x128-tile register bug of round 2)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seconds", [("tools/stress.py", 20), ("tools/stress_grouped.py", 12), ("tools/stress_split.py", 12),
                                          ("tests/quant_stress.py", 25)])   # the last: every quantizer kernel against the oracle, byte for byte
def test_seeded_stress_slice(dev, tool, seconds):
    env = dict(os.environ, STRESS_SEED="12345")
    r = subprocess.run([sys.executable, os.path.join(ROOT, tool), str(seconds)], env=env, capture_output=True, text=True,
                       timeout=300)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-8:])
    assert r.returncode == 0, tail
    assert "0 mismatches" in r.stdout, tail

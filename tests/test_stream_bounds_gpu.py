"""The weight-streaming kernel never reads behind its operands: see tests/bounds_probe.py (a child process, because the
failure mode is a GPU memory fault)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_operands_at_the_end_of_their_allocations():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bounds_probe.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "done" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    cases = [l.split() for l in p.stdout.splitlines() if l.startswith("case")]
    assert len(cases) == 8
    for c in cases:
        assert c[-1] == c[-2], c            # the same bytes as with the operands in the middle of torch's pool

"""Persistent launches of the 256 x 256 tile (round 5; mx_gemm_tile.inc "Persistent launches"): every output bit equals the one-tile
kernels', which the oracle tests hold.  Child processes, because MICROMIX_GEMM_PERSIST is read once per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(persist):
    env = dict(os.environ, MICROMIX_GEMM_PERSIST=persist)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "persist_probe.py")], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0 and "done" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
    return [l for l in p.stdout.splitlines() if l.startswith(("gemm", "act"))]


def test_persistent_launches_are_bit_identical():
    one, per = run("0"), run("1")
    assert len(one) == len(per) == 11
    for a, b in zip(one, per):
        assert a == b, (a, b)

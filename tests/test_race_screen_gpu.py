"""Race screen for the hand-scheduled kernels (counted vmcnt / lgkmcnt, raw s_barrier): they are deterministic, so repeated
launches on the same inputs must be bit-identical (tools/stress_determinism.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hot_kernels_are_deterministic(dev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_determinism.py"), "60"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "TOTAL mismatches 0" in out.stdout

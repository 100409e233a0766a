"""A short run of the randomised parity sweep (scripts/stress_parity.py): ragged and degenerate batches of every
RANSAC / pose / matching leg compared bit for bit with the oracle, small random bundle adjustments within the gates.
(30000 rounds of it ran clean on the MI355X box when it was written; this keeps 150 in the suite.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_parity.py"), "150"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "150 rounds, 0 mismatches" in out.stdout

"""A short run of the randomised parity sweep (scripts/stress_parity.py): ragged and degenerate batches of every
RANSAC / pose / matching leg compared bit for bit with the oracle, small random bundle adjustments (all-free ring scenes,
and aerial scenes with random frozen cameras / points / intrinsics, several intrinsics blocks, GPS rows) within the gates.
(30000 rounds of the first version and 1600 rounds with the masked bundle adjustments, 200 of those, ran clean on the
MI355X box when they were written; this keeps 150 rounds in the suite.)"""
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

"""The C++ host mirror (host/objectsfm.{h,cc}) and the test_sfm driver run end to end on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_test_sfm_driver():
    exe = os.path.join(ROOT, "host", "test_sfm")
    assert os.path.exists(exe), "host/test_sfm not built (run __graft_entry__.build())"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "test_sfm ok" in out.stdout

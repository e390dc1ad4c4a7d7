"""Runs the C++ program that drives the hot path through the C++17 shim (include/dfx_shim.hpp), i.e. the way the
reference's CameraTracker / PhotometricFactor would call it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_cpp_shim_tracker():
    exe = os.path.join(ROOT, "tests", "cpp", "shim_test")
    assert os.path.exists(exe), "tests/cpp/shim_test not built: run __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim_test OK" in out.stdout

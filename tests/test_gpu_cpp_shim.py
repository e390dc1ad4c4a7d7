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


@pytest.mark.gpu
def test_cpp_host_layer():
    """include/dfx_host.hpp (C++17 keyframe store + LinearizeAll over the C ABI): tests/cpp/host_test.cpp, compiled with plain g++."""
    exe = os.path.join(ROOT, "tests", "cpp", "host_test")
    assert os.path.exists(exe), "tests/cpp/host_test not built: run __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_test OK" in out.stdout


@pytest.mark.gpu
def test_reference_callers_compiled_unmodified_against_the_shim():
    """tests/cpp/ref_callers_test.cpp #includes the reference's core/gtsam/photometric_factor.cpp and core/system/camera_tracker.cpp whole and
    unmodified (built by tests/cpp/Makefile in the container that holds /root/reference; the build FAILS if a reference call site stops
    compiling against include/dfx_shim.hpp) and checks linearize() / error() / TrackFrame() against the C ABI bit for bit."""
    exe = os.path.join(ROOT, "tests", "cpp", "ref_callers_test")
    assert os.path.exists(exe), "tests/cpp/ref_callers_test not built: run __graft_entry__.build() where /root/reference exists"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ref_callers_test OK" in out.stdout

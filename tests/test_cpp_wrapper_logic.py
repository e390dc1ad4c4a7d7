"""The wrapper around the C++ test programs (tests/test_gpu_cpp_shim.py::_run) must tell a program that hangs BEFORE its verdict (failure) from one that has
printed its verdict and then does not return from the GPU runtime's teardown (warning) -- checked here with stand-in programs, no GPU."""
import importlib.util
import os
import stat
import sys
import warnings

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("cpp_shim_tests", os.path.join(ROOT, "tests", "test_gpu_cpp_shim.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)


def _script(tmp_path, name, body):
    p = tmp_path / name
    p.write_text("#!" + sys.executable + "\nimport sys, time\n" + body)
    p.chmod(p.stat().st_mode | stat.S_IXUSR)
    return str(p)


def test_verdict_then_clean_exit(tmp_path):
    exe = _script(tmp_path, "ok", "print('x_test OK'); sys.stdout.flush()\n")
    assert "x_test OK" in mod._run(exe, "x_test OK", timeout=30)


def test_nonzero_exit_fails(tmp_path):
    exe = _script(tmp_path, "bad", "print('x_test OK'); sys.exit(3)\n")
    with pytest.raises(AssertionError):
        mod._run(exe, "x_test OK", timeout=30)


def test_hang_after_the_verdict_is_a_warning(tmp_path):
    exe = _script(tmp_path, "late", "print('x_test OK'); sys.stdout.flush(); time.sleep(60)\n")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = mod._run(exe, "x_test OK", timeout=3)
    assert "x_test OK" in out and any("did not exit" in str(x.message) for x in w)


def test_hang_before_the_verdict_fails(tmp_path):
    exe = _script(tmp_path, "early", "print('working'); sys.stdout.flush(); time.sleep(60)\n")
    with pytest.raises(AssertionError, match="hung before its verdict"):
        mod._run(exe, "x_test OK", timeout=3)

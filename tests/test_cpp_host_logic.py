"""Pure host logic of the C++17 layer (include/dfx_host.hpp, dfx_shim.hpp) WITHOUT a GPU: the relinearisation test of
PhotometricFactor::GetJacobiansIfNeeded (photometric_factor.cpp:296-306), the residual rescaling (:275-282), the HessianFactor block
slicing (:105-161) and the item accessors -- tests/cpp/host_logic_test.cpp, plain g++, no device entry point is called."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_logic_without_a_gpu():
    exe = os.path.join(ROOT, "tests", "cpp", "host_logic_test")
    if not os.path.exists(exe):   # built by __graft_entry__.build(); build it here when the tests run first
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "host_logic_test"], stdout=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_logic_test OK" in out.stdout


def test_cpp_exchange_step_two_ranks_over_a_stub_rccl():
    """tests/cpp/comm_test.cpp: dfx_comm_* / dfx_shard_range / dfx_comm_reduce_f32_async / dfx_items_all_gather_async of the C ABI with world
    size 2 (two threads) -- libdfx resolves RCCL at run time, here from the host-memory stand-in tests/cpp/librccl_stub.so (DFX_RCCL_LIB)."""
    d = os.path.join(ROOT, "tests", "cpp")
    if not (os.path.exists(os.path.join(d, "comm_test")) and os.path.exists(os.path.join(d, "librccl_stub.so"))):
        subprocess.check_call(["make", "-C", d, "comm_test", "librccl_stub.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, DFX_RCCL_LIB=os.path.join(d, "librccl_stub.so"))
    out = subprocess.run([os.path.join(d, "comm_test")], capture_output=True, text=True, timeout=60, stdin=subprocess.DEVNULL, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "comm_test OK" in out.stdout


def _rccl_abi_check(*args):
    d = os.path.join(ROOT, "tests", "cpp")
    exe = os.path.join(d, "rccl_abi_check")
    if not os.path.exists(exe) and os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        subprocess.check_call(["make", "-C", d, "rccl_abi_check"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=120, stdin=subprocess.DEVNULL)


def test_hand_declared_rccl_subset_matches_the_real_header_and_library():
    """Pre-flight of the exchange that needs no second GPU (tests/cpp/rccl_abi_check.cpp): the prototypes and enumerator values dfx_comm.cpp
    declares by hand (deepfactors_amd/csrc/dfx_rccl_abi.hpp) are checked against the REAL <rccl/rccl.h> by static_asserts -- the binary exists only
    if they hold -- and every entry point resolves in the real librccl.  (No collective runs here: there is no device.)"""
    out = _rccl_abi_check()
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rccl_abi_check OK" in out.stdout

"""bench.py host logic that can be checked without a GPU: `python bench.py --gpus N` (the driver's single-process invocation) re-launches
itself as N ranks through torch.distributed.run on 127.0.0.1; an external launcher (RANK set) is left alone; the PMC child is never
re-spawned."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("dfx_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_n_without_a_launcher_spawns_n_ranks(monkeypatch):
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(b.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_defaults_are_the_driver_contract():
    b = _bench()
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = b.parse()
    finally:
        sys.argv = old
    assert (a.gpus, a.pairs, a.width, a.height, a.cs) == (1, 128, 640, 480, 32)   # BASELINE.json metric: 640x480, 32-code, N = 1 by default
    assert a.steps > 0 and a.warmup >= 0 and a.schedule == "auto"


def test_world_size_must_match_gpus(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        b.main()
    assert "WORLD_SIZE=4" in str(e.value.code)


def test_help_prints(monkeypatch, capsys):
    """argparse expands help strings with %: a bare percent sign in one of them made `bench.py --help` raise (round 6)."""
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--help"])
    with pytest.raises(SystemExit) as e:
        b.parse()
    assert e.value.code == 0
    assert "--workload" in capsys.readouterr().out

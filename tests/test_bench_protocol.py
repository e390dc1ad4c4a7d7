"""bench.py's measurement protocol (run_protocol: clock ramp in windows, warm-up, timed steps) on two gloo ranks with a fake context and
a fake step that contains a collective: ranks whose kernel times settle at different moments must still issue the same number of steps
and collectives (a divergence would hang the collective inside the step -- the failure mode of an N > 1 run that no 1-GPU box can
show).  The protocol no longer probes schedules: schedule and evaluation mode are the library's defaults (or forced on the command
line), and the context must not be touched."""
import importlib.util
import os
import socket
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("dfx_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FakeCtx:
    """Kernel time model: ramps down over the first `settle` recorded launches, then flat; the dynamic schedule is `dyn_gain` faster."""

    def __init__(self, settle, dyn_gain):
        self.settle, self.dyn_gain = settle, dyn_gain
        self.profiling, self.pending, self.launched, self.dynamic = False, [], 0, False
        self.mode_log = []

    def launch(self):
        self.launched += 1
        t = 1.0 + 0.5 * max(0.0, 1.0 - self.launched / self.settle)
        if self.dynamic:
            t *= self.dyn_gain
        if self.profiling:
            self.pending.append(t)

    def set_profiling(self, on):
        self.profiling, self.pending = bool(on), []

    def profile_read(self):
        n, ms = len(self.pending), sum(self.pending)
        self.pending = []
        return n, ms

    def profile_read_ex(self):
        lo, hi = (min(self.pending), max(self.pending)) if self.pending else (0.0, 0.0)
        n, ms = self.profile_read()
        return n, ms, lo, hi

    def set_schedule(self, mode):
        from deepfactors_amd import _lib
        self.dynamic = mode == _lib.DFX_SCHEDULE_DYNAMIC
        self.mode_log.append(mode)

    def last_schedule_dynamic(self):
        return self.dynamic


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = _bench()
    # rank 0 settles after ~150 launches, rank 1 needs ~500
    ctx = FakeCtx(settle=150 if rank == 0 else 500, dyn_gain=0.98 if rank == 0 else 1.01)
    counters = dict(steps=0, barriers=0)
    tok = torch.zeros(1)

    def step():
        ctx.launch()
        counters["steps"] += 1
        dist.all_reduce(tok)              # the exchange step: hangs if the ranks' step counts ever differ

    def barrier():
        counters["barriers"] += 1
        dist.barrier()

    a = types.SimpleNamespace(schedule="auto", warmup=3, steps=7)
    pr = b.run_protocol(a, dist, torch.device("cpu"), ctx, step, barrier, P=128)
    out[rank] = dict(steps=counters["steps"], barriers=counters["barriers"], ramp=pr["ramp_steps"], sched=a.schedule, modes=len(ctx.mode_log),
                     n_launch=pr["n_launch"], windows=len(pr["hist"]), elapsed=pr["elapsed"], final_dynamic=ctx.dynamic,
                     spread=(pr["kern_min_ms"], pr["kern_max_ms"], pr["kern_ms"]))
    dist.barrier()
    dist.destroy_process_group()


def test_protocol_keeps_two_ranks_in_lockstep():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    assert r0["steps"] == r1["steps"] and r0["barriers"] == r1["barriers"] and r0["ramp"] == r1["ramp"] and r0["windows"] == r1["windows"]
    assert r0["windows"] > 6                                   # rank 1 had not settled after the minimum six windows: both stayed
    assert r0["sched"] == r1["sched"] == "auto" and r0["modes"] == r1["modes"] == 0   # nothing probed, the context's schedule never touched
    assert not r0["final_dynamic"] and not r1["final_dynamic"]
    lo, hi, tot = r0["spread"]
    assert 0 < lo <= tot / 7 <= hi
    assert r0["n_launch"] == r1["n_launch"] == 7               # exactly K timed steps were profiled
    assert r0["elapsed"] == r1["elapsed"] > 0                  # MAX over ranks


def test_protocol_single_process():
    b = _bench()
    ctx = FakeCtx(settle=100, dyn_gain=0.97)
    n = dict(steps=0)

    def step():
        ctx.launch(); n["steps"] += 1
    a = types.SimpleNamespace(schedule="auto", warmup=2, steps=5)
    pr = b.run_protocol(a, None, torch.device("cpu"), ctx, step, lambda: None, P=128)
    assert a.schedule == "auto" and not ctx.dynamic and not ctx.mode_log
    assert pr["n_launch"] == 5 and n["steps"] == pr["ramp_steps"] + 2 + 5 and len(pr["hist"]) >= 6
    assert pr["kern_min_ms"] <= pr["kern_ms"] / 5 <= pr["kern_max_ms"]
    a2 = types.SimpleNamespace(schedule="static", warmup=0, steps=4)
    ctx2 = FakeCtx(settle=10, dyn_gain=1.0)
    pr2 = b.run_protocol(a2, None, torch.device("cpu"), ctx2, ctx2.launch, lambda: None, P=128)
    assert pr2["n_launch"] == 4 and not ctx2.mode_log

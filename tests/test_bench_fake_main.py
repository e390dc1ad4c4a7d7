"""bench.py's main() end to end WITHOUT a GPU: the device layer is replaced by fakes (a context with a synthetic kernel clock, an aligner
whose items come from the CPU oracle and whose assembly is the torch formulation of deepfactors_amd.dist), `cuda` devices map to the
CPU and RCCL to gloo.  What this pins is everything around the kernels that no 1-GPU box can show: the N = 2 path (shards, the
overlapped reduce through two system buffers, the checksum of the exchanged system on rank 0, ONE JSON line from rank 0 only) and the
contract fields of the line.  The oracle is used here as the stand-in device, in a test -- never by bench.py itself."""
import contextlib
import importlib.util
import io
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--pairs", "48", "--width", "64", "--height", "48", "--cs", "16", "--steps", "3", "--warmup", "1", "--no-configs", "--no-traffic", "--no-cpu-baseline"]


class FakeCtx:
    handle = None   # the C-ABI exchange (deepfactors_amd.dist.Comm) passes it through: a null context = no stream, which the host-memory RCCL stand-in ignores anyway

    def __init__(self, device=None, stream="torch"):
        self.profiling, self.pending, self.dynamic, self.mfma = False, [], False, 0

    def launch(self):
        if self.profiling:
            self.pending.append(1.0 if not self.dynamic else 0.97)

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

    def set_tail_stream(self, stream):
        self.tail = stream

    def tail_join(self):
        self.joins = getattr(self, "joins", 0) + 1

    def set_schedule(self, mode):
        from deepfactors_amd import _lib
        self.dynamic = mode == _lib.DFX_SCHEDULE_DYNAMIC

    def last_schedule_dynamic(self):
        return self.dynamic

    def set_mfma_mode(self, mode):
        self.mfma = mode

    def sync(self):
        pass

    def alloc_image(self, w, h, elems_per_px=1):   # library-owned valid0 maps of bench.build_pairs: plain host memory here
        import torch
        return torch.zeros((h, w), dtype=torch.float32)

    def last_mfma_mode(self):
        from deepfactors_amd import _lib
        return _lib.DFX_MFMA_BF16X3 if self.mfma == _lib.DFX_MFMA_AUTO else self.mfma


class FakeAligner:
    """Items from the CPU oracle (computed once per batch), assembly through NormalEquations.assemble (torch, CPU).  Only the FIRST
    aligner of a process (main's timed workload, whose results main() checks) evaluates anything; the aligners of the secondary
    measurements just tick the fake kernel clock."""
    created = 0

    def __init__(self, params=None, code_size=32, ctx=None):
        self.CS, self.ctx, self._bytes = int(code_size), ctx, None
        self.real = FakeAligner.created == 0
        FakeAligner.created += 1

    def make_pairs(self, pairs):
        return list(pairs)

    def RunStep(self, *args):
        self.ctx.launch()

    def LinearizeBatch(self, arr, prx, codes, items):
        assert len(prx) == len(arr) == len(codes)
        self.ctx.launch()

    def _items(self, arr):
        if not self.real:
            return None
        if self._bytes is None:
            from deepfactors_amd._lib import item_inliers_offset, item_jtj_len, item_size
            from oracle import dfx_oracle as orc
            orc.build()
            NP = 12 + self.CS
            raw = np.zeros((len(arr), item_size(NP)), np.uint8)
            for k, p in enumerate(arr):
                r = orc.sfm_step(p["pose0"], p["pose1"], p["cam"], p["img0"].numpy(), p["img1"].numpy(), p["dpt0"].numpy(), p["prx0_jac"].numpy(), p["grad1"].numpy())
                f = raw[k, : (item_jtj_len(NP) + NP + 1) * 4].view(np.float32)
                f[: item_jtj_len(NP)] = r.JtJ
                f[item_jtj_len(NP): item_jtj_len(NP) + NP] = r.Jtr
                f[item_jtj_len(NP) + NP] = r.residual
                raw[k, item_inliers_offset(NP):].view(np.uint64)[0] = r.inliers
            self._bytes = torch.from_numpy(raw.reshape(-1))
        return self._bytes

    def EvaluateErrorBatch(self, arr, items=None):
        self.ctx.launch()

    def RunStepBatchAsync(self, arr, items):
        it = self._items(arr)
        if it is not None:
            items.copy_(it)
        self.ctx.launch()

    def RunStepBatchAssembleAsync(self, arr, items, neq, first_pair, fused=True):
        from deepfactors_amd import item_size
        self.RunStepBatchAsync(arr, items)
        neq.assemble(items, first_pair, len(arr), item_size(12 + self.CS))

    @staticmethod
    def items_from_bytes(raw, cs):
        from deepfactors_amd.aligners import SfmAligner
        return SfmAligner.items_from_bytes(raw, cs)


class FakeSE3:
    def __init__(self, ctx=None):
        self.ctx = ctx

    def make_pairs(self, pairs):
        return list(pairs)

    def RunStepBatch(self, arr, items=None):
        self.ctx.launch()


class FakeEvent:
    def __init__(self, enable_timing=False):
        pass

    def record(self):
        pass

    def elapsed_time(self, other):
        return 1.0


class FakeStream:
    def __init__(self, device=None, priority=0):
        pass


def _run_main(argv, out_path):
    """bench.main() with the device layer faked; stdout of the call goes to out_path."""
    import deepfactors_amd
    import torch.distributed as tdist
    spec = importlib.util.spec_from_file_location("dfx_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # the single-process measurements that drive the tracker, the geometric factor and the host solver are device code end to end (covered on the
    # GPU by tests/test_gpu_tracker.py, test_gpu_sparse_geometric.py, test_gpu_window.py): here only their place in main()'s control flow
    b.tracker_and_geometric_configs = lambda *a, **k: {"configs0_se3_tracker_3level": {"ms_per_frame": 1.0}, "configs2_sparse_geometric_500pts": {"round_ms": 1.0},
                                                       "configs2_gauss_newton_round_16kf_120pairs": {"total_ms": 1.0}}
    b.gauss_newton_round = lambda *a, **k: {"total_ms": 1.0}
    b.cpu_baseline_se3 = lambda *a, **k: {"ms_per_frame": 1.0, "cores": 1}
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)   # the stand-in device runs hundreds of tiny torch ops per second: intra-op threads only add wake-up latency (58 ms vs 0.3 ms per assembly)
    real_device, real_init = torch.device, tdist.init_process_group
    from deepfactors_amd import synth
    real_make_pair = synth.make_pair
    FakeAligner.created = 0

    def small_make_pair(w=640, h=480, cs=32, **kw):   # the secondary configurations ask for 640x480 ... 1280x960 pairs: content is irrelevant to the fakes
        return real_make_pair(min(w, 64), min(h, 48), cs, **kw)
    patches = [(torch.cuda, "is_available", lambda: True), (torch.cuda, "set_device", lambda d: None), (torch.cuda, "synchronize", lambda *a: None),
               (torch, "device", lambda *a, **k: real_device("cpu")), (deepfactors_amd, "Context", FakeCtx), (deepfactors_amd, "SfmAligner", FakeAligner),
               (tdist, "init_process_group", lambda backend, rank, world_size, device_id=None: real_init("gloo", rank=rank, world_size=world_size)),
               (torch.cuda, "Stream", FakeStream), (torch.cuda, "stream", lambda s: contextlib.nullcontext()), (torch.cuda, "empty_cache", lambda: None),
               (torch.cuda, "set_stream", lambda s: None),
               (torch.cuda, "Event", FakeEvent), (deepfactors_amd, "SE3Aligner", FakeSE3), (deepfactors_amd, "UpdateDepthBatch", lambda *a, **k: None),
               (deepfactors_amd, "BuildPyramids", lambda *a, **k: None), (deepfactors_amd, "make_pyramids", lambda *a, **k: [None]), (deepfactors_amd, "GaussianBlurDown", lambda *a, **k: None),
               (deepfactors_amd, "SobelGradients", lambda *a, **k: None),
               (synth, "make_pair", small_make_pair)]
    saved = [(o, n, getattr(o, n)) for o, n, _ in patches]
    old_argv = sys.argv
    buf = io.StringIO()
    try:
        for o, n, v in patches:
            setattr(o, n, v)
        sys.argv = ["bench.py"] + argv
        with contextlib.redirect_stdout(buf):
            b.main()
    finally:
        sys.argv = old_argv
        torch.set_num_threads(n_threads)
        for o, n, v in saved:
            setattr(o, n, v)
    with open(out_path, "w") as fh:
        fh.write(buf.getvalue())


def _check_line(txt, world):
    lines = [l for l in txt.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["metric"].startswith("keyframe-pair residual+Jacobian evals/sec") and d["unit"] == "pair-evals/s"
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - world * 48 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert d["config"]["pairs_per_gpu"] == 48 and "workload" in d["config"] and ("RCCL reduce" in d["config"]["workload"]) == (world > 1)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["launches"] == 3 and r["traffic"] is None and "traffic_source" in r and "mfma" in r and "schedule" in r
    assert "schedule_probe" not in d and len(d["ramp_kernel_us"]) >= 6
    assert 0 < r["kernel_us_min"] <= r["kernel_us"] <= r["kernel_us_max"] and "library default" in r["mfma"] + r["schedule"]
    # the tail runs in order unless --deferred-tail -- or the C-ABI exchange moves it (and the collective) to the tail stream for N > 1
    assert ("second stream" in d["config"]["workload"]) == (world > 1 and d["config"]["exchange"] == "cabi")
    assert d["config"]["exchange"] == ("none" if world == 1 else d["config"]["exchange"])
    return d


def test_main_single_process(tmp_path, monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    out = tmp_path / "n1.txt"
    _run_main(ARGS, str(out))
    _check_line(out.read_text(), 1)


STUB = os.path.join(ROOT, "tests", "cpp", "librccl_stub_shm.so")


def _stub():
    """The host-memory, multi-process stand-in for RCCL (tests/cpp/rccl_stub_shm.cpp), built on demand."""
    if not os.path.exists(STUB):
        import subprocess
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "cpp", "rccl_stub_shm.cpp"), "-o", STUB, "-lrt", "-lpthread"])
    return STUB


def _rank(rank, world, port, outdir, exchange, rccl=STUB):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DFX_RCCL_LIB=rccl)
    _run_main(["--gpus", str(world), "--window", "--exchange", exchange] + ARGS, os.path.join(outdir, f"rank{rank}.txt"))


@pytest.mark.parametrize("exchange", ["cabi", "torch"])
def test_main_two_ranks_over_gloo(tmp_path, exchange):
    """N = 2.  exchange = cabi (bench.py's default): the SHIPPED collectives -- dfx_comm_get_unique_id on rank 0, the 128 bytes handed round over the
    process group, dfx_comm_create, dfx_comm_reduce_f32_async per step (deepfactors_amd/csrc/dfx_comm.cpp) -- run for real, over a host-memory stand-in
    for librccl loaded through DFX_RCCL_LIB; main()'s checksum of the exchanged system on rank 0 asserts that the sum arrived.  exchange = torch: the
    torch.distributed collectives (gloo here) on the same buffers, pipelined over two systems."""
    _stub()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank, args=(2, port, str(tmp_path), exchange), nprocs=2, join=True)
    assert (tmp_path / "rank1.txt").read_text().strip() == ""          # rank 0 alone prints the line
    d = _check_line((tmp_path / "rank0.txt").read_text(), 2)
    assert d["config"]["exchange"] == exchange and (("C ABI" in d["config"]["workload"]) == (exchange == "cabi"))
    assert "not collected for N > 1" in d["roofline"]["traffic_source"] and "cpu_baseline" not in d
    w = d["configs"]["configs3_window64"]                               # --window: BASELINE configs[3] sharded over the two ranks
    assert w["keyframes"] == 64 and w["pairs"] == 1024 and w["pairs_per_rank"] == 512 and w["evals_per_s"] > 0


def test_main_four_ranks_over_the_c_abi_exchange(tmp_path):
    """N = 4 through the shipped collectives (host-memory RCCL stand-in): the graph spans 4 x 48 pairs, every rank assembles its shard into the common
    system, rank 0's checksum of the reduced system equals the checksum of all ranks' items (asserted inside main())."""
    _stub()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank, args=(4, port, str(tmp_path), "cabi"), nprocs=4, join=True)
    for r in (1, 2, 3):
        assert (tmp_path / f"rank{r}.txt").read_text().strip() == ""
    d = _check_line((tmp_path / "rank0.txt").read_text(), 4)
    assert d["config"]["exchange"] == "cabi" and d["n_gpus"] == 4
    assert d["configs"]["configs3_window64"]["pairs_per_rank"] == 256


def test_main_eight_ranks_over_the_c_abi_exchange(tmp_path):
    """N = 8: the node size the driver's scaling run ends at."""
    _stub()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank, args=(8, port, str(tmp_path), "cabi"), nprocs=8, join=True)
    d = _check_line((tmp_path / "rank0.txt").read_text(), 8)
    assert d["config"]["exchange"] == "cabi" and d["n_gpus"] == 8 and d["configs"]["configs3_window64"]["pairs_per_rank"] == 128


def test_main_two_ranks_fall_back_together_when_the_communicator_cannot_be_created(tmp_path):
    """A rank whose dfx_comm_create fails (here: an RCCL library that cannot be loaded) must not leave the others waiting in a collective: the ranks agree over
    the process group, all of them exchange through torch.distributed, and the line says so."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_rank, args=(2, port, str(tmp_path), "cabi", "/nonexistent/librccl.so"), nprocs=2, join=True)
    d = json.loads((tmp_path / "rank0.txt").read_text().strip().splitlines()[-1])
    assert d["config"]["exchange"].startswith("torch (dfx_comm_create failed") and d["n_gpus"] == 2 and d["value"] > 0
    assert "C ABI" not in d["config"]["workload"]


def test_main_with_the_secondary_configurations(tmp_path, monkeypatch):
    """The default command's extra measurements (configs[1] single pair and pyramid, configs[4] on both evaluation modes, the configs[2]
    relinearisation round, the headline batch on the bf16 split): every code path of secondary_configs / mode_kernel_us runs and lands in
    the line."""
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    out = tmp_path / "cfg.txt"
    _run_main([a for a in ARGS if a != "--no-configs"], str(out))
    d = _check_line(out.read_text(), 1)
    c = d["configs"]
    assert set(c) >= {"headline_workload_other_mode", "configs1_single_pair_blocking", "configs1_pyramid3_128pairs", "configs4_1280x960_cs64", "configs2_linearize_16kf_120pairs",
                      "update_depth_batch_64kf", "se3_step_batch_128pairs", "sfm_error_batch_128pairs", "configs3_window64",
                      "configs0_se3_tracker_3level", "configs2_sparse_geometric_500pts", "configs2_gauss_newton_round_16kf_120pairs",
                      "headline_truth", "headline_unrelated", "configs2_linearize_16kf_240pairs_both_directions"}
    # the default workload is the perturbed one (round 6); the two it did not run are measured beside it, each with its inlier fraction
    assert d["config"]["poses"] == "perturbed" and "headline_perturbed" not in c and "perturbed" in d["config"]["workload"]
    for k in ("headline_truth", "headline_unrelated"):
        assert c[k]["kernel_us"] > 0 and 0 <= c[k]["mean_inliers_frac"] <= 1
    assert c["configs2_linearize_16kf_240pairs_both_directions"]["pairs"] == 240
    assert "gauss_newton_round" in c["configs3_window64"]
    assert c["headline_workload_other_mode"]["kernel_us"] > 0 and "error" not in c["headline_workload_other_mode"] and "fp32 fmaf chain" in c["headline_workload_other_mode"]["mfma"]
    assert c["configs4_1280x960_cs64"]["f32_chain"]["kernel_us"] > 0 and "error" not in c["configs4_1280x960_cs64"]["f32_chain"]
    assert c["configs4_1280x960_cs64"]["frac"] > 0 and len(c["configs1_pyramid3_128pairs"]["level_by_level_kernel_us"]) == 3 and c["configs1_pyramid3_128pairs"]["one_launch_kernel_us"] > 0
    for k in ("update_depth_batch_64kf", "se3_step_batch_128pairs", "sfm_error_batch_128pairs"):
        assert c[k]["us"] > 0 and c[k]["frac"] > 0


def test_main_forced_modes(tmp_path, monkeypatch):
    """--schedule static --mfma bf16x3 --window on one process: no probe, the line says which evaluation mode ran."""
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    out = tmp_path / "forced.txt"
    _run_main(ARGS + ["--schedule", "static", "--mfma", "bf16x3", "--window", "--deferred-tail"], str(out))
    d = json.loads(out.read_text().strip())
    assert "second stream" in d["config"]["workload"]
    assert "bf16" in d["roofline"]["mfma"] and "forced by --mfma" in d["roofline"]["mfma"] and "static" in d["roofline"]["schedule"] and "forced" in d["roofline"]["schedule"]
    assert d["configs"]["configs3_window64"]["pairs_per_rank"] == 1024

"""Keyframe store: host-side logic on CPU (gloo replication, TUM export), pyramid construction on the GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepfactors_amd.keyframe import Keyframe, KeyframeMap
    kf = Keyframe(2, 32, 24, 16, device="cpu")
    if rank == 1:   # rank 1 "built" the keyframe
        g = torch.Generator().manual_seed(7)
        for t in kf.tensors():
            t.copy_(torch.rand(t.shape, generator=g))
        kf.id, kf.code = 42, np.arange(16, dtype=np.float32) / 10
        kf.pose_wk = np.array([0.1, 0.2, 0.3, 0.9, 1, 2, 3], np.float32)
    m = KeyframeMap()
    m.Broadcast(kf, dist, src=1)
    out[rank] = (m.Ids(), float(sum(float(t.sum()) for t in kf.tensors())), kf.code.tolist(), kf.pose_wk.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_keyframe_broadcast_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][0] == out[1][0] == [42]
    assert abs(out[0][1] - out[1][1]) < 1e-6 and out[0][1] > 0
    assert out[0][2] == out[1][2] and out[0][3] == out[1][3]


def test_tum_export(tmp_path):
    from deepfactors_amd.keyframe import save_trajectory_tum
    p = tmp_path / "traj.txt"
    save_trajectory_tum(str(p), [0.5, 1.5], [np.array([0, 0, 0, 1, 1, 2, 3.0]), np.array([0, 0.1, 0, 0.99, 4, 5, 6.0])])
    lines = p.read_text().strip().split("\n")
    assert lines[0].split() == ["0.500000", "1.000000", "2.000000", "3.000000", "0.000000", "0.000000", "0.000000", "1.000000"]
    assert len(lines) == 2 and lines[1].startswith("1.500000 4.000000")


@pytest.mark.gpu
def test_keyframe_build_matches_oracle(dfx, oracle):
    """Mapper::BuildKeyframe's data flow (mapper.cpp:933-1000) on the device: pyramid, gradients, depth per level, depth
    gradient -- each level against the oracle."""
    from deepfactors_amd import synth
    from deepfactors_amd.keyframe import Keyframe
    w, h, cs, L = 128, 96, 32, 3
    n = synth.to_numpy(synth.make_pair(w, h, cs, seed=55))
    kf = Keyframe(L, w, h, cs, device="cuda")
    kf.FillPyramids(torch.from_numpy(n["img0"]))
    # decoder outputs per level: level 0 from the synthetic decoder, coarser levels by subsampling (stand-in for the network)
    prx = [n["prx_orig"][:: 2 ** i, :: 2 ** i] for i in range(L)]
    jac = [n["prx_jac"].reshape(h, w, cs)[:: 2 ** i, :: 2 ** i].reshape(h >> i, (w >> i) * cs) for i in range(L)]
    kf.SetDecoderOutputs([np.ascontiguousarray(p) for p in prx], [np.zeros_like(p) for p in prx], [np.ascontiguousarray(j) for j in jac])
    kf.UpdateDepthMaps(n["code"], 2.0)
    img = n["img0"]
    for i in range(L):
        if i > 0:
            img = oracle.blur_down(img)
        assert np.abs(kf.pyr_img[i].cpu().numpy() - img).max() <= 1e-6
        assert np.array_equal(kf.pyr_grad[i].cpu().numpy(), oracle.sobel(kf.pyr_img[i].cpu().numpy()))
        d_ref = oracle.update_depth(n["code"], np.ascontiguousarray(prx[i]), np.ascontiguousarray(jac[i]), 2.0)
        assert np.abs(kf.pyr_dpt[i].cpu().numpy() - d_ref).max() <= 2e-6 * float(((2.0 + d_ref) ** 2 / 2.0).max())
        assert float(kf.pyr_vld[i].min()) == 1.0
    assert np.array_equal(kf.dpt_grad.cpu().numpy(), oracle.sobel(kf.pyr_dpt[0].cpu().numpy()))
    assert kf.nbytes() > 0 and kf.IsKeyframe() and kf.Name() == "kf0"

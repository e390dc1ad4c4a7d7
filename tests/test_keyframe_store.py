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


def test_network_config_loader(tmp_path):
    """decoder_network.cpp:231-325 on a config with the reference's keys."""
    import json
    from deepfactors_amd import DfxError
    from deepfactors_amd.keyframe import LoadJsonNetworkConfig
    cfg = {"graph_path": "net.pb", "input_width": 256, "input_height": 192, "pyramid_levels": 4, "code_size": 32, "grayscale": True,
           "avg_dpt": 2.0, "input_names": {"image": "input/image:0", "code": "input/code:0"},
           "output_names": {"depth_est": ["d0:0", "d1:0"], "depth_stdev": ["s0:0", "s1:0"], "depth_jac": ["j0:0", "j1:0"]},
           "camera": {"fx": 200.5, "fy": 201.5, "u0": 127.5, "v0": 95.5}}
    p = tmp_path / "scannet256.cfg"
    p.write_text(json.dumps(cfg))
    c = LoadJsonNetworkConfig(str(p))
    assert c.graph_path == str(tmp_path / "net.pb") and (c.input_width, c.input_height, c.pyramid_levels, c.code_size) == (256, 192, 4, 32)
    assert c.input_image_name == "input/image" and c.depth_jac_names == ["j0", "j1"] and not c.depth_pred
    assert np.allclose(c.camera_array(), [200.5, 201.5, 127.5, 95.5, 256, 192])
    del cfg["code_size"]
    p.write_text(json.dumps(cfg))
    with pytest.raises(DfxError):
        LoadJsonNetworkConfig(str(p))
    with pytest.raises(DfxError):
        LoadJsonNetworkConfig(str(tmp_path / "missing.cfg"))


def test_save_results_formats(tmp_path):
    """SaveResults / SaveKeyframes (deepfactors.cpp:539-594): TUM trajectory, depth x 5000 as 16-bit PNG, intrinsics.txt."""
    from PIL import Image
    from deepfactors_amd.keyframe import Keyframe, save_results
    kf = Keyframe(1, 16, 12, 16, device="cpu")
    kf.timestamp, kf.id = 12.5, 3
    kf.pose_wk = np.array([0, 0, 0, 1, 0.1, 0.2, 0.3], np.float32)
    d = np.linspace(0.0, 14.0, 16 * 12, dtype=np.float32).reshape(12, 16)   # 14 m x 5000 saturates uint16
    kf.pyr_dpt[0].copy_(torch.from_numpy(d))
    kf.color_img = (np.arange(16 * 12 * 3) % 251).astype(np.uint8).reshape(12, 16, 3)
    save_results(str(tmp_path), [kf], [100.0, 101.0, 8.0, 6.0, 16, 12])
    got = np.array(Image.open(tmp_path / "keyframes" / "12.500000_dpt.png"))
    assert got.dtype in (np.uint16, np.int32) and np.array_equal(got.astype(np.int64), np.clip(np.rint(d.astype(np.float64) * 5000), 0, 65535).astype(np.int64))
    assert np.array_equal(np.array(Image.open(tmp_path / "keyframes" / "12.500000_rgb.png")), kf.color_img)
    assert (tmp_path / "keyframes" / "intrinsics.txt").read_text() == "100 101 8 6 16 12"
    assert (tmp_path / "trajectory.txt").read_text().split() == ["12.500000", "0.100000", "0.200000", "0.300000", "0.000000", "0.000000", "0.000000", "1.000000"]

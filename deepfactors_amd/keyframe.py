"""Device-resident frame / keyframe store.

Mirrors the data model of the reference (``df::Frame`` mapping/frame.h:36-120, ``df::Keyframe`` mapping/keyframe.h:34-100,
``Mapper::BuildKeyframe`` mapping/mapper.cpp:921-1004) with the buffers that feed the alignment path, but without the
reference's ``SyncedBufferPyramid`` (cuda/synced_pyramid.h): that class lazily mirrors every level between CPU and GPU
with dirty flags (and LOG(FATAL)s on divergence) because the decoder network and the sparse factors run on the CPU.
On an MI355X everything this path reads fits in HBM thousands of times over (a 640x480x32 keyframe pyramid is ~60 MB of
288 GB), so a pyramid here is simply a list of device tensors, level 0 first; host copies are explicit (``.cpu()``).

The decoder network itself (TensorFlow, core/network/decoder_network.cpp) is out of scope: its three outputs per
level -- zero-code proximity, log-uncertainty and the code Jacobian in ``[H][W*CS]`` layout -- enter through
``Keyframe.SetDecoderOutputs``.
"""
import numpy as np
import torch

from . import aligners as _al


def _alloc_pyr(levels, w, h, device, ch=None):
    out = []
    for i in range(levels):
        shape = (h >> i, w >> i) if ch is None else (h >> i, w >> i, ch)
        out.append(torch.zeros(shape, dtype=torch.float32, device=device))
    return out


class Frame:
    """``df::Frame<float>``: image pyramid + gradient pyramid on the device, pose and id on the host."""

    def __init__(self, pyrlevels, w, h, device="cuda", ctx=None):
        self.width, self.height, self.levels = int(w), int(h), int(pyrlevels)
        self.device = torch.device(device)
        self.ctx = ctx
        self.pyr_img = _alloc_pyr(pyrlevels, w, h, self.device)
        self.pyr_grad = _alloc_pyr(pyrlevels, w, h, self.device, 2)
        self.pose_wk = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
        self.id = 0
        self.timestamp = 0.0
        self.marginalized = False

    def Name(self):
        return f"fr{self.id}"

    def IsKeyframe(self):
        return False

    def FillPyramids(self, img, pyrlevels=None, blocking=True):
        """frame.h:80-94: level 0 = img, level i = GaussianBlurDown(level i-1); Sobel gradient on every level.  BLOCKING by default, like the
        reference (its image operators each end in cudaDeviceSynchronize, launch_utils.h:26-32): the pyramids are complete -- readable from any
        stream, launch errors raised here -- on return.  blocking=False only enqueues on the context's stream (FillPyramidsBatch always does)."""
        n = self.levels if pyrlevels is None else int(pyrlevels)
        img = torch.as_tensor(img, dtype=torch.float32)
        if tuple(img.shape) != (self.height, self.width):
            raise _al.DfxError(-1, f"image is {tuple(img.shape)}, frame is {(self.height, self.width)}")
        self.pyr_img[0].copy_(img)   # H2D (or D2D) upload on the current stream
        # one enqueue: a launch per level, each reading its level once (Sobel + blur-down from one LDS tile); same bits as the per-level operators
        _al.BuildPyramids([self.pyr_img[:n]], [self.pyr_grad[:n]], self.ctx, blocking=blocking)

    @staticmethod
    def FillPyramidsBatch(frames, imgs=None, pyrlevels=None):
        """FillPyramids of several frames in ONE enqueue (one launch per pyramid level over all frames).  `imgs[k]` (optional) is uploaded into
        frame k's level 0 first.  ENQUEUE ONLY: the pyramids are complete when the context's stream reaches this point."""
        frames = list(frames)
        n = frames[0].levels if pyrlevels is None else int(pyrlevels)
        if imgs is not None:
            for f, im in zip(frames, imgs):
                f.pyr_img[0].copy_(torch.as_tensor(im, dtype=torch.float32))
        _al.BuildPyramids([f.pyr_img[:n] for f in frames], [f.pyr_grad[:n] for f in frames], frames[0].ctx)

    def tensors(self):
        return list(self.pyr_img) + list(self.pyr_grad)


class Keyframe(Frame):
    """``df::Keyframe<float>``: adds depth / valid / stdev / zero-code proximity / code-Jacobian pyramids, the level-0
    depth gradient (used by the sparse geometric factor) and the code."""

    def __init__(self, pyrlevels, w, h, cs, device="cuda", ctx=None):
        super().__init__(pyrlevels, w, h, device, ctx)
        self.cs = int(cs)
        self.pyr_dpt = _alloc_pyr(pyrlevels, w, h, self.device)
        self.pyr_vld = [torch.ones_like(t) for t in self.pyr_dpt]          # mapper.cpp:937 fillBuffer(pyr_vld, 1.0f)
        self.pyr_stdev = _alloc_pyr(pyrlevels, w, h, self.device)
        self.pyr_prx_orig = _alloc_pyr(pyrlevels, w, h, self.device)
        self.pyr_jac = [torch.zeros(((h >> i), (w >> i) * cs), dtype=torch.float32, device=self.device) for i in range(pyrlevels)]
        self.dpt_grad = torch.zeros((h, w, 2), dtype=torch.float32, device=self.device)
        self.code = np.zeros(cs, np.float32)

    def Name(self):
        return f"kf{self.id}"

    def IsKeyframe(self):
        return True

    def SetDecoderOutputs(self, pyr_prx_orig, pyr_stdev, pyr_jac):
        """The decoder's per-level outputs (decoder_network.cpp:126-136); pyr_jac[i] is [H_i][W_i*CS] (keyframe.h:52)."""
        for i in range(self.levels):
            self.pyr_prx_orig[i].copy_(torch.as_tensor(pyr_prx_orig[i], dtype=torch.float32))
            self.pyr_stdev[i].copy_(torch.as_tensor(pyr_stdev[i], dtype=torch.float32))
            self.pyr_jac[i].copy_(torch.as_tensor(pyr_jac[i], dtype=torch.float32).reshape(self.pyr_jac[i].shape))

    def UpdateDepthMaps(self, code=None, avg_dpt=2.0, use_geometric=True):
        """mapper.cpp:984-1000 / 881-887: depth of every level from the code, then the level-0 depth gradient."""
        if code is not None:
            self.code = np.asarray(code, np.float32).reshape(self.cs).copy()
        for i in range(self.levels):
            _al.UpdateDepth(self.code, self.pyr_prx_orig[i], self.pyr_jac[i], avg_dpt, self.pyr_dpt[i], self.ctx)
        if use_geometric:
            _al.SobelGradients(self.pyr_dpt[0], self.dpt_grad, self.ctx)

    def tensors(self):
        return (super().tensors() + list(self.pyr_dpt) + list(self.pyr_vld) + list(self.pyr_stdev) + list(self.pyr_prx_orig)
                + list(self.pyr_jac) + [self.dpt_grad])

    def nbytes(self):
        return sum(t.numel() * 4 for t in self.tensors())


class KeyframeMap:
    """Keyframes by id (core/mapping/keyframe_map.h, data side only) + the multi-GPU replication of SURVEY 8e: every rank
    holds every keyframe pyramid (64 keyframes x ~60 MB << 288 GB), a new keyframe is broadcast once from the rank that
    built it, and only pair lists are sharded afterwards."""

    def __init__(self):
        self.keyframes = {}

    def Add(self, kf):
        self.keyframes[kf.id] = kf
        return kf

    def Get(self, kf_id):
        return self.keyframes[kf_id]

    def Ids(self):
        return sorted(self.keyframes)

    def Broadcast(self, kf, dist, src):
        """One-time replication of a keyframe's buffers from rank `src` (RCCL broadcast over xGMI; gloo in the CPU tests)."""
        for t in kf.tensors():
            dist.broadcast(t, src)
        meta = torch.zeros(8 + kf.cs, dtype=torch.float64, device=kf.device)
        if dist.get_rank() == src:
            meta[:7] = torch.as_tensor(np.asarray(kf.pose_wk, np.float64))
            meta[7] = float(kf.id)
            meta[8:] = torch.as_tensor(np.asarray(kf.code, np.float64))
        dist.broadcast(meta, src)
        m = meta.cpu().numpy()
        kf.pose_wk, kf.id, kf.code = m[:7].astype(np.float32), int(m[7]), m[8:].astype(np.float32)
        self.keyframes[kf.id] = kf
        return kf


def save_trajectory_tum(path, timestamps, poses_wk):
    """TUM trajectory export, one line `timestamp tx ty tz qx qy qz qw` per pose (deepfactors.cpp:541-560)."""
    with open(path, "w") as f:
        for ts, p in zip(timestamps, poses_wk):
            p = np.asarray(p, np.float64)
            f.write(f"{ts:.6f} {p[4]:.6f} {p[5]:.6f} {p[6]:.6f} {p[0]:.6f} {p[1]:.6f} {p[2]:.6f} {p[3]:.6f}\n")


# ---- formats at the edges of the path ---------------------------------------------------------------------
import json as _json
import os as _os
import struct as _struct
import zlib as _zlib
from dataclasses import dataclass, field
from typing import List


@dataclass
class NetworkConfig:
    """``DecoderNetwork::NetworkConfig`` (core/network/decoder_network.h): what the mapper needs to know about the decoder whose
    outputs feed ``Keyframe.SetDecoderOutputs`` -- input size, pyramid depth, code size, avg_dpt, camera, tensor names."""
    graph_path: str = ""
    input_width: int = 0
    input_height: int = 0
    pyramid_levels: int = 0
    code_size: int = 0
    grayscale: bool = True
    avg_dpt: float = 2.0
    input_image_name: str = ""
    input_code_name: str = ""
    depth_est_names: List[str] = field(default_factory=list)
    depth_std_names: List[str] = field(default_factory=list)
    depth_jac_names: List[str] = field(default_factory=list)
    depth_pred: bool = False
    depth_pred_names: List[str] = field(default_factory=list)
    code_pred_name: str = ""
    camera: dict = field(default_factory=dict)   # fx fy u0 v0 (of the network's input size)

    def camera_array(self):
        """[fx, fy, u0, v0, w, h] as the aligners take it."""
        c = self.camera
        return np.array([c["fx"], c["fy"], c["u0"], c["v0"], self.input_width, self.input_height], np.float32)


def LoadJsonNetworkConfig(cfgpath):
    """decoder_network.cpp:231-325: same keys, same checks (a missing key is an error), tensor names cut after the last ':',
    a relative graph_path is resolved against the directory of the cfg file."""
    try:
        with open(cfgpath) as fh:
            root = _json.load(fh)
    except OSError as e:
        raise _al.DfxError(-1, f"Could not load network config: {cfgpath} ({e})")

    def need(node, key, what="network config"):
        if key not in node:
            raise _al.DfxError(-1, f"{what}: missing key '{key}' in {cfgpath}")
        return node[key]

    cut = lambda s: s[: s.rfind(":")] if ":" in s else s
    names = lambda v: [cut(x) for x in v]
    cfg = NetworkConfig()
    gp = need(root, "graph_path")
    cfg.graph_path = gp if gp.startswith("/") else _os.path.join(_os.path.dirname(cfgpath), gp)
    cfg.input_width, cfg.input_height = int(need(root, "input_width")), int(need(root, "input_height"))
    cfg.pyramid_levels, cfg.code_size = int(need(root, "pyramid_levels")), int(need(root, "code_size"))
    cfg.grayscale, cfg.avg_dpt = bool(need(root, "grayscale")), float(need(root, "avg_dpt"))
    inn = need(root, "input_names")
    cfg.input_image_name, cfg.input_code_name = cut(need(inn, "image", "input_names")), cut(need(inn, "code", "input_names"))
    on = need(root, "output_names")
    for key, attr in (("depth_est", "depth_est_names"), ("depth_stdev", "depth_std_names"), ("depth_jac", "depth_jac_names")):
        v = need(on, key, "output_names")
        if not isinstance(v, list):
            raise _al.DfxError(-1, f"output_names.{key} must be an array in {cfgpath}")
        setattr(cfg, attr, names(v))
    if root.get("depth_pred"):
        cfg.depth_pred = True
        cfg.depth_pred_names = names(need(on, "depth_pred", "output_names"))
        cfg.code_pred_name = names(need(on, "code_pred", "output_names"))[0]
    cam = need(root, "camera")
    cfg.camera = {k: float(need(cam, k, "camera")) for k in ("fx", "fy", "u0", "v0")}
    return cfg


def write_png(path, img):
    """Minimal PNG encoder (zlib only): uint8 [H][W] / [H][W][3] or uint16 [H][W] (big-endian samples, as PNG stores them)."""
    a = np.ascontiguousarray(img)
    if a.dtype == np.uint16 and a.ndim == 2:
        depth, ctype, raw = 16, 0, a.astype(">u2").tobytes()
    elif a.dtype == np.uint8 and a.ndim == 2:
        depth, ctype, raw = 8, 0, a.tobytes()
    elif a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3:
        depth, ctype, raw = 8, 2, a.tobytes()
    else:
        raise _al.DfxError(-1, f"write_png: unsupported array {a.dtype} {a.shape}")
    h, w = a.shape[:2]
    stride = len(raw) // h
    scan = b"".join(b"\x00" + raw[y * stride:(y + 1) * stride] for y in range(h))   # filter type 0 on every scanline

    def chunk(tag, data):
        return _struct.pack(">I", len(data)) + tag + data + _struct.pack(">I", _zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", _struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
                + chunk(b"IDAT", _zlib.compress(scan, 6)) + chunk(b"IEND", b""))


def save_keyframes(directory, keyframes, cam):
    """``DeepFactors::SaveKeyframes`` (core/deepfactors.cpp:539-570): per keyframe `<timestamp>_dpt.png` = level-0 depth x 5000 as
    16-bit PNG (saturating, like cv::Mat::convertTo) and `<timestamp>_rgb.png` when the keyframe carries `color_img`
    (uint8 [H][W][3]); `intrinsics.txt` = "fx fy u0 v0 w h"."""
    kdir = _os.path.join(directory, "keyframes")
    _os.makedirs(kdir, exist_ok=True)
    for kf in keyframes:
        ts = f"{kf.timestamp:.6f}"   # std::to_string(double)
        d = kf.pyr_dpt[0].detach().cpu().numpy().astype(np.float64) * 5000.0
        write_png(_os.path.join(kdir, ts + "_dpt.png"), np.clip(np.rint(d), 0, 65535).astype(np.uint16))
        col = getattr(kf, "color_img", None)
        if col is not None:
            write_png(_os.path.join(kdir, ts + "_rgb.png"), np.asarray(col, np.uint8))
    c = np.asarray(cam, np.float64)
    with open(_os.path.join(kdir, "intrinsics.txt"), "w") as f:
        f.write(" ".join(f"{v:g}" for v in c[:6]))


def save_results(directory, keyframes, cam):
    """``DeepFactors::SaveResults`` (core/deepfactors.cpp:573-594): trajectory.txt (TUM, keyframe poses) + SaveKeyframes."""
    _os.makedirs(directory, exist_ok=True)
    save_trajectory_tum(_os.path.join(directory, "trajectory.txt"), [kf.timestamp for kf in keyframes], [kf.pose_wk for kf in keyframes])
    save_keyframes(directory, keyframes, cam)

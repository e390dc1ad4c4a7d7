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

    def FillPyramids(self, img, pyrlevels=None):
        """frame.h:80-94: level 0 = img, level i = GaussianBlurDown(level i-1); Sobel gradient on every level."""
        n = self.levels if pyrlevels is None else int(pyrlevels)
        img = torch.as_tensor(img, dtype=torch.float32)
        if tuple(img.shape) != (self.height, self.width):
            raise _al.DfxError(-1, f"image is {tuple(img.shape)}, frame is {(self.height, self.width)}")
        self.pyr_img[0].copy_(img)   # H2D (or D2D) upload on the current stream
        for i in range(n):
            if i > 0:
                _al.GaussianBlurDown(self.pyr_img[i - 1], self.pyr_img[i], self.ctx)
            _al.SobelGradients(self.pyr_img[i], self.pyr_grad[i], self.ctx)

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

"""Deterministic synthetic keyframe pairs for tests, smoke() and bench.py.

There is no network for datasets or decoder weights, so every input is generated: a band-limited analytic texture,
a smooth depth map, a GT relative motion, and a smooth random linear decoder (``prx_orig``, ``prx_jac``) standing in
for the TensorFlow network's outputs (mapping/keyframe.h:46-56: ``pyr_prx_orig``, ``pyr_jac [H][W*CS]``).
Because the texture is analytic, ``img0`` is photometrically consistent with ``(dpt_true, T10_true)`` exactly:
``img0(x) = tex(warp(x))`` and ``img1 = tex`` on the pixel grid.

All random draws come from ``numpy.random.default_rng(seed)`` (few hundred scalars: sinusoid parameters, codes);
fields are evaluated with torch in float64 on the requested device and rounded to float32 once.
"""
import math

import numpy as np
import torch

SCENENET_TAN_X = 0.5773502691896257   # tests/testing_utils.h:34-40  GetSceneNetCam
SCENENET_TAN_Y = 0.41421356237309503


def scenenet_cam(w, h):
    """``df::GetSceneNetCam<float>(w, h)`` (tests/testing_utils.h:34-40): integer w/2, h/2 like the reference."""
    fx = np.float32((w // 2) / SCENENET_TAN_X)
    fy = np.float32((h // 2) / SCENENET_TAN_Y)
    return np.array([fx, fy, w // 2, h // 2, w, h], np.float32)


def camera_pyramid(cam, levels):
    """``df::CameraPyramid`` (camera_pyramid.h:35-48) + ``ResizeViewport`` (pinhole_camera_impl.h:126-136), float32."""
    cams = [np.asarray(cam, np.float32).copy()]
    for _ in range(1, levels):
        c = cams[-1].copy()
        nw, nh = float(int(c[4]) // 2), float(int(c[5]) // 2)
        xr, yr = np.float32(nw / c[4]), np.float32(nh / c[5])
        c[0] *= xr; c[1] *= yr; c[2] *= xr; c[3] *= yr
        c[4], c[5] = nw, nh
        cams.append(c.astype(np.float32))
    return cams


def so3_exp(w):
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-9:
        return np.eye(3) + K
    return np.eye(3) + math.sin(th) / th * K + (1 - math.cos(th)) / th ** 2 * (K @ K)


def R_to_quat(R):
    tr = np.trace(R)
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s, (R[2, 1] - R[1, 2]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s, (R[0, 2] - R[2, 0]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s, (R[1, 0] - R[0, 1]) / s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def quat_to_R(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def pose_qt(R, t):
    return np.concatenate([R_to_quat(np.asarray(R, np.float64)), np.asarray(t, np.float64)]).astype(np.float32)


IDENTITY = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)


def _sinusoid_field(rng, n, lam_lo, lam_hi):
    lam = rng.uniform(lam_lo, lam_hi, n)
    ang = rng.uniform(0, 2 * math.pi, n)
    return dict(kx=2 * math.pi / lam * np.cos(ang), ky=2 * math.pi / lam * np.sin(ang), ph=rng.uniform(0, 2 * math.pi, n),
                am=rng.uniform(0.5, 1.0, n))


def _eval_field(f, u, v):
    out = torch.zeros_like(u)
    for kx, ky, ph, am in zip(f["kx"], f["ky"], f["ph"], f["am"]):
        out = out + am * torch.sin(kx * u + ky * v + ph)
    return out


def sobel_torch(img):
    """(gx, gy)/8 with clamped borders -- same taps as SobelGradients (cu_image_proc.cpp:34-92); data prep only."""
    p = torch.nn.functional.pad(img[None, None].double(), (1, 1, 1, 1), mode="replicate")[0, 0]
    gx = (-p[:-2, :-2] + p[:-2, 2:] - 2 * p[1:-1, :-2] + 2 * p[1:-1, 2:] - p[2:, :-2] + p[2:, 2:]) / 8
    gy = (-p[:-2, :-2] - 2 * p[:-2, 1:-1] - p[:-2, 2:] + p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) / 8
    return torch.stack([gx, gy], -1).float().contiguous()


def make_pair(w=640, h=480, cs=32, seed=0xDF02, device="cpu", avg_dpt=2.0, motion_scale=1.0, code_sigma=0.3,
              jac_amp=0.05, with_decoder=True):
    """One keyframe->frame pair in the layout SfmAligner::RunStep consumes.

    Returns a dict of float32 torch tensors on `device` (img0, img1, dpt0, grad1, prx_orig, prx_jac [H][W*cs], std0,
    valid0) plus numpy metadata (cam, pose0, pose1, code, avg_dpt, pose10_true).  `dpt0` is decode(code) exactly as
    UpdateDepth defines it (computed in float64, rounded once)."""
    rng = np.random.default_rng(seed)
    dev = torch.device(device)
    cam = scenenet_cam(w, h)
    fx, fy, u0, v0 = [float(c) for c in cam[:4]]
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float64, device=dev), torch.arange(w, dtype=torch.float64, device=dev),
                            indexing="ij")
    sx = w / 640.0   # feature sizes scale with resolution so every size sees the same scene

    # depth: tilted plane + bumps, clamp [0.8, 6]
    d = 2.5 + 0.3 * ((xs - u0) / w) - 0.3 * ((ys - v0) / h)
    for _ in range(3):
        cx, cy = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h
        sg, am = rng.uniform(40, 90) * sx, rng.uniform(-0.4, 0.4)
        d = d + am * torch.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sg * sg))
    d = d.clamp(0.8, 6.0)

    out = dict(cam=cam, avg_dpt=float(avg_dpt), w=w, h=h, cs=cs, seed=seed)
    if with_decoder:
        # linear decoder: prx = prx_orig + jac . code ; basis fields smooth, |j|_inf ~ jac_amp
        code = rng.normal(0.0, code_sigma, cs)
        prx_true = avg_dpt / (avg_dpt + d)
        jac = torch.empty((h, w, cs), dtype=torch.float32, device=dev)
        jdotc = torch.zeros_like(d)
        for k in range(cs):
            f = _sinusoid_field(rng, 8, 40 * sx, 400 * sx)
            fk = _eval_field(f, xs, ys)
            fk = (fk * (jac_amp / float(np.sum(f["am"])))).float()
            jac[:, :, k] = fk
            jdotc = jdotc + fk.double() * float(np.float32(code[k]))
        prx_orig = (prx_true - jdotc).float()
        # depth the aligner sees = decode(code) in the reference's formula
        prx = prx_orig.double() + jdotc
        d = avg_dpt / prx - avg_dpt
        out.update(code=code.astype(np.float32), prx_orig=prx_orig.contiguous(), prx_jac=jac.reshape(h, w * cs).contiguous())
    dpt0 = d.float()

    # GT motion (SURVEY 8d cfg 1/2): small twist, scaled
    tw_t = np.array([0.04, -0.03, 0.02]) * motion_scale
    tw_w = np.array([0.01, -0.015, 0.008]) * motion_scale
    R10, t10 = so3_exp(tw_w), tw_t
    # pose0 = identity, pose1 = T10^-1  (pose_10 = pose1^-1 * pose0)
    pose0 = IDENTITY.copy()
    pose1 = pose_qt(R10.T, -R10.T @ t10)

    tex = _sinusoid_field(rng, 24, 16 * sx, 160 * sx)
    tnorm = float(np.sum(tex["am"]))

    def texture(u, v):
        return 0.5 + 0.5 * _eval_field(tex, u, v) / tnorm

    img1 = texture(xs, ys)
    dd = dpt0.double()
    X = (xs - u0) / fx * dd
    Y = (ys - v0) / fy * dd
    Z = dd
    Rt = torch.tensor(R10, dtype=torch.float64, device=dev)
    qx = Rt[0, 0] * X + Rt[0, 1] * Y + Rt[0, 2] * Z + t10[0]
    qy = Rt[1, 0] * X + Rt[1, 1] * Y + Rt[1, 2] * Z + t10[1]
    qz = Rt[2, 0] * X + Rt[2, 1] * Y + Rt[2, 2] * Z + t10[2]
    img0 = texture(fx * qx / qz + u0, fy * qy / qz + v0)

    img1f = img1.float().contiguous()
    out.update(img0=img0.float().contiguous(), img1=img1f, dpt0=dpt0.contiguous(), grad1=sobel_torch(img1f),
               std0=torch.zeros((h, w), dtype=torch.float32, device=dev), valid0=torch.zeros((h, w), dtype=torch.float32, device=dev),
               pose0=pose0, pose1=pose1, pose10_true=pose_qt(R10, t10))
    return out


def to_numpy(pair):
    """Host copy of every tensor in a pair dict (for the CPU oracle)."""
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in pair.items()}


def to_device(pair, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in pair.items()}

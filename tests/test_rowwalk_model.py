"""Executable specification (numpy, fp32) of the per-pixel arithmetic of the round-4 pixel reductions -- `row_walk`, `se3_step_body`,
`sfm_error_body` in deepfactors_amd/csrc/dfx_misc_kernels.hip and `derive_fast_geo` in dfx_kernels.hpp: fast geometry (fused
multiply-adds, one reciprocal, validity as a margin in homogeneous coordinates) with the reference-order evaluation as the fall-back
inside the ambiguity band.  Checked against the oracle (checker only): the inlier set is EXACTLY the reference-order one, the sums are
within the stated tolerance, and the band is both rare and sufficient (no pixel outside it is classified differently)."""
from types import SimpleNamespace

import numpy as np
import pytest

from deepfactors_amd import synth
from oracle import dfx_oracle as orc
from helpers import assert_item_close

f32 = np.float32
U24 = 5.9604644775390625e-08


def derive_fast(R, t, cam, W, H):
    """dfx_kernels.hpp derive_fast_cam + derive_fast_geo (double arithmetic, rounded once)."""
    fx, fy, u0, v0, w, h = [float(c) for c in cam]
    R = np.asarray(R, np.float64).reshape(9)
    t = np.asarray(t, np.float64)
    rxmax = f32(max(abs(0.0 - u0), abs((W - 1) - u0)) / abs(fx) * 1.000001)
    rymax = f32(max(abs(0.0 - v0), abs((H - 1) - v0)) / abs(fy) * 1.000001)
    G = abs(fx) + abs(fy) + 2.0 * (w + h) + 2.0 * (abs(u0) + abs(v0))
    gscale = f32(32.0 * U24 * G * 1.000001)
    cu, cv = u0 - 0.5 * w, v0 - 0.5 * h
    g = dict(cu=f32(cu), cv=f32(cv), hw=f32(0.5 * w - 1.0), hh=f32(0.5 * h - 1.0))
    g["KR"] = np.array([fx * R[0] + cu * R[6], fx * R[1] + cu * R[7], fx * R[2] + cu * R[8],
                        fy * R[3] + cv * R[6], fy * R[4] + cv * R[7], fy * R[5] + cv * R[8], R[6], R[7], R[8]]).astype(f32)
    g["Kt"] = np.array([fx * t[0] + cu * t[2], fy * t[1] + cv * t[2], t[2]]).astype(f32)
    fw, fh = np.floor(0.5 * w), np.floor(0.5 * h)
    g["fcx"], g["fcy"], g["icx"], g["icy"] = f32(0.5 * w - fw), f32(0.5 * h - fh), int(fw), int(fh)
    g["du"], g["dv"] = f32(-cu), f32(-cv)
    rho = max(abs(R[3 * i]) * float(rxmax) + abs(R[3 * i + 1]) * float(rymax) + abs(R[3 * i + 2]) for i in range(3))
    tau = max(abs(t[i]) for i in range(3))
    g["e1"], g["e2"] = f32(float(gscale) * rho * 1.000001), f32(float(gscale) * tau * 1.000001)
    return g


def fma(a, b, c):
    """fp32 fused multiply-add: exact product and sum in double (53 bits hold a 24 x 24-bit product; the one rounding of the
    double sum before the rounding to fp32 can differ from a true fma by a double-rounding in ~2^-29 of the cases: irrelevant here)."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def reference_order(R, t, cam, rx, ry, d):
    """find_correspondence_ray (dfx_device.hpp) = warping.h:204-241 in fp32, separate multiplies and adds, IEEE division."""
    fx, fy, u0, v0, w, h = [f32(c) for c in cam]
    R = R.astype(f32); t = t.astype(f32)
    px, py, pz = rx * d, ry * d, d
    v = [(R[3 * i] * px + R[3 * i + 1] * py) + R[3 * i + 2] * pz for i in range(3)]
    qx, qy, qz = v[0] + t[0], v[1] + t[1], v[2] + t[2]
    with np.errstate(all="ignore"):
        u = (fx * qx) / qz + u0
        vv = (fy * qy) / qz + v0
        iz = f32(1.0) / qz
    valid = (qz > 0) & (u >= 1) & (u < w - f32(1)) & (vv >= 1) & (vv < h - f32(1))
    return u, vv, iz, valid


def row_walk_model(qt, cam, img0, img1, dpt0, grad1, huber_delta, want_grad):
    H, W = img0.shape
    Rd = synth.quat_to_R(np.asarray(qt[:4], np.float64))
    R = Rd.astype(f32).reshape(9)            # what fill_simple stores
    t = np.asarray(qt[4:], f32)
    g = derive_fast(R.astype(np.float64), t.astype(np.float64), cam, W, H)
    fx, fy, u0, v0 = [f32(c) for c in cam[:4]]
    # the host's ray table: (x - u0) / fx in fp32 (dfx_api.cpp ray_table)
    rx = ((np.arange(W, dtype=f32) - u0) / fx)[None, :].repeat(H, 0)
    ry = ((np.arange(H, dtype=f32) - v0) / fy)[:, None].repeat(W, 1)
    d = dpt0.astype(f32)
    # (K) R ray: the column part is loop-invariant per lane, the row part one fma per component (row_walk: cx, cy, cz)
    M = R if want_grad else g["KR"]
    rr = [fma(M[3 * i + 1], ry, fma(M[3 * i], rx, M[3 * i + 2])) for i in range(3)]
    if want_grad:
        vx, vy, vz = rr[0] * d, rr[1] * d, rr[2] * d
        Z = vz + t[2]
        X = fma(fx, vx + t[0], g["cu"] * Z)
        Y = fma(fy, vy + t[1], g["cv"] * Z)
    else:
        Kt = g["Kt"]
        X, Y, Z = fma(rr[0], d, Kt[0]), fma(rr[1], d, Kt[1]), fma(rr[2], d, Kt[2])
    with np.errstate(all="ignore"):
        iz = (f32(1.0) / Z).astype(f32)
        mu, mv = fma(-g["hw"], Z, np.abs(X)), fma(-g["hh"], Z, np.abs(Y))
        E = fma(g["e1"], np.abs(d), g["e2"])
        valid = (mu < -E) & (mv < -E)
        amb = np.abs(np.fmax(mu, mv)) < E
        tu, tv = fma(X, iz, g["fcx"]), fma(Y, iz, g["fcy"])
        U, V = fma(X, iz, g["du"]), fma(Y, iz, g["dv"])
    ue, ve, ize, valid_e = reference_order(R, t, cam, rx, ry, d)
    fast_valid = valid.copy()
    valid = np.where(amb, valid_e, valid)
    tu = np.where(amb, ue - f32(g["icx"]), tu); tv = np.where(amb, ve - f32(g["icy"]), tv)
    U = np.where(amb, ue - u0, U); V = np.where(amb, ve - v0, V); iz = np.where(amb, ize, iz)
    stats = dict(amb=int(amb.sum()), outside_band_mismatch=int(((fast_valid != valid_e) & ~amb).sum()), inliers=int(valid.sum()),
                 exact_inliers=int(valid_e.sum()), set_equal=bool(np.array_equal(valid, valid_e)))
    ys, xs = np.nonzero(valid)
    tu, tv, U, V, iz = tu[ys, xs], tv[ys, xs], U[ys, xs], V[ys, xs], iz[ys, xs]
    # the snap: a tap coordinate less than 2^-13 pixel below an integer is that integer (biased floor, weight clamped at 0)
    SNAP = f32(2.0 ** -13)
    fu, fv = np.minimum(np.floor(tu + SNAP), f32(W - 2 - g["icx"])), np.minimum(np.floor(tv + SNAP), f32(H - 2 - g["icy"]))   # never past the last cell
    ax, ay = np.maximum(tu - fu, f32(0)), np.maximum(tv - fv, f32(0))
    ix, iy = fu.astype(np.int64) + g["icx"], fv.astype(np.int64) + g["icy"]
    stats["ix_minus_x"] = (ix - xs)
    stats["iy_minus_y"] = (iy - ys)
    assert ix.min() >= 0 and ix.max() + 1 <= W - 1 and iy.min() >= 0 and iy.max() + 1 <= H - 1, "tap out of range"
    lerp = lambda a, b, w: fma(w, b - a, a)
    samp = lerp(lerp(img1[iy, ix], img1[iy, ix + 1], ax), lerp(img1[iy + 1, ix], img1[iy + 1, ix + 1], ax), ay)
    r = img0[ys, xs] - samp
    aa = np.abs(r)
    with np.errstate(all="ignore"):
        wgt = np.where(aa <= f32(huber_delta), f32(1), np.sqrt(f32(huber_delta) * (f32(2) * aa - f32(huber_delta))) / aa).astype(f32)
    r = r * wgt
    if not want_grad:
        return float(np.sum(r.astype(np.float64) ** 2)), stats
    gx = lerp(lerp(grad1[iy, ix, 0], grad1[iy, ix + 1, 0], ax), lerp(grad1[iy + 1, ix, 0], grad1[iy + 1, ix + 1, 0], ax), ay)
    gy = lerp(lerp(grad1[iy, ix, 1], grad1[iy, ix + 1, 1], ax), lerp(grad1[iy + 1, ix, 1], grad1[iy + 1, ix + 1, 1], ax), ay)
    wz = wgt * iz
    J = np.zeros((6, len(r)), f32)
    J[0] = -(gx * fx) * wz
    J[1] = -(gy * fy) * wz
    J[2] = fma(gx, U, gy * V) * wz
    vx, vy, vz = vx[ys, xs], vy[ys, xs], vz[ys, xs]
    J[3] = vy * J[2] - vz * J[1]
    J[4] = vz * J[0] - vx * J[2]
    J[5] = vx * J[1] - vy * J[0]
    J64 = J.astype(np.float64)
    JtJ = J64 @ J64.T
    res = SimpleNamespace(JtJ=JtJ[np.triu_indices(6)], Jtr=J64 @ r.astype(np.float64), residual=float(np.sum(r.astype(np.float64) ** 2)),
                          inliers=int(valid.sum()))
    return res, stats


CASES = [(160, 120, 1.0, 5), (320, 240, 1.0, 6), (96, 72, 3.0, 7), (81, 61, 1.0, 8)]


@pytest.mark.parametrize("w,h,motion,seed", CASES)
def test_fast_geometry_keeps_the_reference_inlier_set_and_the_sums(w, h, motion, seed):
    n = synth.to_numpy(synth.make_pair(w, h, 16, seed=seed, with_decoder=False, motion_scale=motion))
    for qt in (synth.IDENTITY.copy(), n["pose10_true"]):
        ref = orc.se3_step(qt, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], 0.1)
        got, st = row_walk_model(qt, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], 0.1, True)
        assert st["set_equal"] and st["outside_band_mismatch"] == 0, st
        assert got.inliers == ref.inliers, (got.inliers, ref.inliers)
        # the band is rare: a few pixels per frame -- except at the exact identity, where the border columns / rows map ONTO the border
        assert st["amb"] <= (2 * (w + h) if np.array_equal(qt, synth.IDENTITY) else 8), st
        assert_item_close(got, ref, w, h, what="row-walk model, SE3")
        # EvaluateError through the K-folded rows
        p0 = synth.IDENTITY.copy()
        p1 = _inverse(qt)
        eref = orc.sfm_error(p0, p1, n["cam"], n["img0"], n["img1"], n["dpt0"], 0.1)
        egot, est = row_walk_model(qt, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], 0.1, False)
        assert est["set_equal"] and est["outside_band_mismatch"] == 0 and est["inliers"] == eref[1], (est, eref)
        assert abs(egot - eref[0]) <= 1e-4 * abs(eref[0]) + 1e-6


def test_degenerate_depths_fall_back_or_drop_out():
    """zero / negative / NaN / huge depths: no pixel may be classified valid by the fast test where the reference order says no."""
    n = synth.to_numpy(synth.make_pair(96, 72, 16, seed=11, with_decoder=False))
    d = n["dpt0"].copy()
    d[::7, ::5] = 0.0
    d[1::7, ::5] = -1.0
    d[2::7, ::5] = np.nan
    d[3::7, ::5] = 1e30
    d[4::7, ::5] = 1e-30
    d[5::7, ::5] = np.inf
    for qt in (synth.IDENTITY.copy(), n["pose10_true"]):
        _, st = row_walk_model(qt, n["cam"], n["img0"], n["img1"], d, n["grad1"], 0.1, False)
        assert st["set_equal"] and st["outside_band_mismatch"] == 0, st


def _inverse(qt):
    R = synth.quat_to_R(np.asarray(qt[:4], np.float64))
    return synth.pose_qt(R.T, -R.T @ np.asarray(qt[4:], np.float64))


def test_the_snap_makes_the_identity_warp_lane_contiguous():
    """At the exact identity u = x + rounding noise: floor(u) would be x or x - 1 at random from pixel to pixel (the dword tap loads of the kernels
    would lose their contiguity); with the snap every inlier taps (x, y) itself -- and the sums still agree with the oracle, for which u = x."""
    w, h = 160, 120
    p = synth.make_pair(w, h, 16, seed=11, device="cpu", with_decoder=False)
    n = synth.to_numpy(p)
    ident = np.asarray(synth.IDENTITY, np.float32)
    got, st = row_walk_model(ident, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], 0.1, True)
    assert st["set_equal"]
    assert np.all(st["ix_minus_x"] == 0) and np.all(st["iy_minus_y"] == 0)
    # without the snap the same arithmetic flips: some pixels land on x - 1 (this is what the snap is for)
    cam = n["cam"]
    Rd = synth.quat_to_R(np.asarray(ident[:4], np.float64))
    g = derive_fast(Rd.reshape(9), np.zeros(3), cam, w, h)
    fx, fy, u0, v0 = [f32(c) for c in cam[:4]]
    rx = ((np.arange(w, dtype=f32) - u0) / fx)[None, :].repeat(h, 0)
    ry = ((np.arange(h, dtype=f32) - v0) / fy)[:, None].repeat(w, 1)
    d = n["dpt0"].astype(f32)
    R = Rd.astype(f32).reshape(9)
    rr = [fma(R[3 * i + 1], ry, fma(R[3 * i], rx, R[3 * i + 2])) for i in range(3)]
    Z = rr[2] * d
    X, Y = fma(fx, rr[0] * d, g["cu"] * Z), fma(fy, rr[1] * d, g["cv"] * Z)
    iz = (f32(1.0) / Z).astype(f32)
    tu = fma(X, iz, g["fcx"])
    flips = np.floor(tu).astype(np.int64) + g["icx"] - np.arange(w)[None, :]
    assert (flips[2:-2, 2:-2] == -1).any() and (flips[2:-2, 2:-2] == 0).any()
    ref = orc.se3_step(ident, n["cam"], n["img0"], n["img1"], n["dpt0"], n["grad1"], 0.1)
    assert got.inliers == ref.inliers
    assert_item_close(got, ref, w, h, what="identity, snapped taps")

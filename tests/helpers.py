"""Shared comparison helpers for the parity tests (tolerances per SURVEY.md section 8c)."""
import numpy as np

# Stated fp32 tolerance of the HIP path against the fp64-accumulating oracle: every JtJ / Jtr / residual entry
# within REL * max|JtJ| (relative to the scale of its block) or ABS, whichever is larger; inlier counts equal up
# to FLIP * W*H boundary flips.
REL = 1e-4
ABS = 1e-5
FLIP = 1e-5


def assert_item_close(got, ref, w, h, rel=REL, what="item"):
    flips = abs(int(got.inliers) - int(ref.inliers))
    assert flips <= max(1, int(FLIP * w * h)), f"{what}: inliers {got.inliers} vs oracle {ref.inliers}"
    # a flipped boundary pixel moves sums by at most one pixel's contribution; widen by that share
    slack = 1.0 + 4.0 * flips
    sj = max(float(np.abs(ref.JtJ).max()), 1e-30)
    dj = float(np.abs(np.asarray(got.JtJ, np.float64) - np.asarray(ref.JtJ, np.float64)).max())
    assert dj <= max(rel * sj * slack, ABS), f"{what}: JtJ max err {dj:.3e} vs scale {sj:.3e}"
    sr = max(float(np.abs(ref.Jtr).max()), 1e-30)
    # Jtr entries are bounded by sqrt(JtJ_ii * residual); use that scale so tiny gradients at a minimum still compare
    scale_r = max(sr, float(np.sqrt(sj * max(ref.residual, 0.0))))
    dr = float(np.abs(np.asarray(got.Jtr, np.float64) - np.asarray(ref.Jtr, np.float64)).max())
    assert dr <= max(rel * scale_r * slack, ABS), f"{what}: Jtr max err {dr:.3e} vs scale {scale_r:.3e}"
    assert abs(got.residual - ref.residual) <= max(rel * abs(ref.residual) * slack, ABS), \
        f"{what}: residual {got.residual} vs {ref.residual}"

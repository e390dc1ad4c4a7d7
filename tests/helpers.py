"""Shared comparison helpers for the parity tests (tolerances per SURVEY.md section 8c, applied PER ENTRY).

Every entry of the normal equations is compared at its own Cauchy-Schwarz scale, never at the scale of the largest entry of
the whole matrix:

    |JtJ_ij - ref_ij|  <=  REL * sqrt(ref_ii * ref_jj)              (|sum J_i J_j| <= sqrt(sum J_i^2 * sum J_j^2))
    |Jtr_i  - ref_i |  <=  REL * sqrt(ref_ii * sum r^2)  +  R_ULPS * eps32 * sqrt(ref_ii * N)
    |res    - ref   |  <=  REL * res  +  2 * R_ULPS * eps32 * sqrt(res * N)  +  N * (R_ULPS * eps32)^2

so the pose-pose (1e6), pose-code (1e2) and code-code (1e0) blocks of the 44 x 44 SfmAligner system -- the six G blocks
GTSAM receives (photometric_factor.cpp:135-161) -- and the code gradient are each pinned to REL of THEIR OWN magnitude.  The
R_ULPS terms are the only absolute floors: a residual r = img0 - bilinear(img1) of unit-range images carries a few ulp(1) of
rounding whatever its size, which bounds what any fp32 evaluation of Jtr / sum r^2 can reproduce when r ~ 0 (N = inliers).

`block_errors` returns the same comparison as numbers, per block, for the reports under profiles/ and bench.py's
`parity_blocks`.
"""
import numpy as np

# Stated fp32 tolerance of the HIP path against the fp64-accumulating oracle (per entry, see above); inlier counts equal up
# to FLIP * W*H boundary flips.
REL = 1e-4
FLIP = 1e-5
R_ULPS = 4.0
EPS32 = float(np.finfo(np.float32).eps)
TINY = 1e-30


def blocks_of(np_):
    """Variable blocks of an item: SE3Aligner (6), SfmAligner (6 + 6 + CS), DepthAligner (CS)."""
    if np_ == 6:
        return {"pose": slice(0, 6)}
    if np_ in (16, 32, 64):   # the supported code sizes; an SfmAligner item is 12 + CS = 28 / 44 / 76
        return {"code": slice(0, np_)}
    assert np_ > 12, np_
    return {"pose0": slice(0, 6), "pose1": slice(6, 12), "code": slice(12, np_)}


def dense(packed, np_):
    """SquareUpperTriangularMatrix::toDenseMatrix() (row-major upper triangle)."""
    M = np.zeros((np_, np_), np.float64)
    M[np.triu_indices(np_)] = np.asarray(packed, np.float64)
    return M + np.triu(M, 1).T


def _n(item):
    return len(np.asarray(item.Jtr))


def block_errors(got, ref):
    """{block name: {"cs": max_ij |d_ij| / sqrt(ref_ii ref_jj), "blk": max|d| / max|ref block|, "scale": max|ref block|}}
    for the blocks G_ab (a <= b) of JtJ and g_a of Jtr (Jtr normalised by sqrt(ref_ii * sum r^2))."""
    np_ = _n(ref)
    G, R = dense(got.JtJ, np_), dense(ref.JtJ, np_)
    d = np.sqrt(np.maximum(np.diag(R), 0.0))
    cs_scale = np.outer(d, d) + TINY
    res = max(float(ref.residual), 0.0)
    inl = max(int(ref.inliers), 1)
    r_scale = d * np.sqrt(res) + R_ULPS * EPS32 * d * np.sqrt(inl) + TINY
    dg = np.abs(np.asarray(got.Jtr, np.float64) - np.asarray(ref.Jtr, np.float64))
    out = {}
    names = list(blocks_of(np_).items())
    for a, (na, sa) in enumerate(names):
        for nb, sb in names[a:]:
            D = np.abs(G[sa, sb] - R[sa, sb])
            m = float(np.abs(R[sa, sb]).max())
            out[f"JtJ[{na},{nb}]"] = dict(cs=float((D / cs_scale[sa, sb]).max()), blk=float(D.max() / (m + TINY)), scale=m)
        m = float(np.abs(np.asarray(ref.Jtr, np.float64)[sa]).max())
        out[f"Jtr[{na}]"] = dict(cs=float((dg[sa] / r_scale[sa]).max()), blk=float(dg[sa].max() / (m + TINY)), scale=m)
    return out


def format_block_errors(errs):
    return "  ".join(f"{k} cs={v['cs']:.1e} blk={v['blk']:.1e} (|max|={v['scale']:.3g})" for k, v in errs.items())


def assert_item_close(got, ref, w, h, rel=REL, what="item"):
    flips = abs(int(got.inliers) - int(ref.inliers))
    assert flips <= max(1, int(FLIP * w * h)), f"{what}: inliers {got.inliers} vs oracle {ref.inliers}"
    # a flipped boundary pixel moves sums by at most one pixel's contribution; widen by that share
    slack = 1.0 + 4.0 * flips
    errs = block_errors(got, ref)
    bad = {k: v for k, v in errs.items() if not v["cs"] <= rel * slack}
    assert not bad, f"{what}: per-entry Cauchy-Schwarz error above {rel * slack:.1e}: {format_block_errors(bad)}"
    res = max(float(ref.residual), 0.0)
    inl = max(int(ref.inliers), 1)
    tol = rel * res * slack + 2.0 * R_ULPS * EPS32 * np.sqrt(res * inl) + inl * (R_ULPS * EPS32) ** 2
    assert abs(got.residual - ref.residual) <= tol, f"{what}: residual {got.residual} vs {ref.residual} (tol {tol:.3e})"
    return errs


class Item:
    """Plain holder with the fields of JTJJrReductionItem, for sums / slices built inside a test."""

    def __init__(self, JtJ, Jtr, residual, inliers):
        self.JtJ, self.Jtr = np.asarray(JtJ, np.float64), np.asarray(Jtr, np.float64)
        self.residual, self.inliers = float(residual), int(inliers)


def item_sum(a, b):
    return Item(np.asarray(a.JtJ, np.float64) + np.asarray(b.JtJ, np.float64), np.asarray(a.Jtr, np.float64) + np.asarray(b.Jtr, np.float64),
                float(a.residual) + float(b.residual), int(a.inliers) + int(b.inliers))


def assert_blocks_below(got, ref, bound, what="item"):
    """Every block's per-entry Cauchy-Schwarz error below `bound` (quality claims tighter than the REL tolerance)."""
    errs = block_errors(got, ref)
    bad = {k: v for k, v in errs.items() if not v["cs"] < bound}
    assert not bad, f"{what}: per-entry Cauchy-Schwarz error not below {bound:.1e}: {format_block_errors(bad)}"
    return errs


def hessian_blocks(item):
    """The six G blocks and three g vectors PhotometricFactor hands to gtsam::HessianFactor (photometric_factor.cpp:135-161):
    G11 G12 G13 G22 G23 G33 of double(JtJ), g = -double(Jtr)."""
    np_ = _n(item)
    M = dense(item.JtJ, np_)
    s = list(blocks_of(np_).values())
    Gs = [M[s[a], s[b]] for a in range(3) for b in range(a, 3)]
    g = -np.asarray(item.Jtr, np.float64)
    return Gs, [g[x] for x in s]

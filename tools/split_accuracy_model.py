#!/usr/bin/env python3
"""Numerical model (numpy, CPU) of the matrix-core evaluation schemes for Z = sum_px z z^T on the step kernel's kind of data: which accuracy does
each buy?  z rows are drawn like the kernel's z-space (a few pose rows of magnitude ~1e2, code rows of ~1e-1, 2 % of the pixels masked),
the reference value is the fp64 sum.
  chain   : fp32 products, sequential fp32 accumulation over the pixels (what v_mfma_f32_16x16x4_f32 computes)
  bf16x3  : exact three-way bf16 split, products hh + hm + mh + hl + lh + mm in fp32 accumulation (the library's default)
  bf16x2  : two bf16 pieces (16 bits), hh + hm + mh
  f16x2   : two fp16 pieces (22 bits), hh + hl + lh -- the candidate of DESIGN.md section 7 -- with the rows scaled by powers of two into fp16's range
Accumulation inside an MFMA is modelled as exact per 32-pixel group, fp32 between the groups of a wave (40 chunks), double between the waves."""
import numpy as np


def bf16_round(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def gram_groups(a, b, group=32, per_partial=2560):
    """sum_px a b^T as the kernel forms it: exact inside an MFMA's group of `group` pixels, fp32 between the groups of one wave's share
    (`per_partial` pixels: 40 chunks), the waves' partial sums in double"""
    n = a.shape[0]
    total = np.zeros((a.shape[1], b.shape[1]), np.float64)
    for p0 in range(0, n, per_partial):
        acc = np.zeros((a.shape[1], b.shape[1]), np.float32)
        for g in range(p0, min(n, p0 + per_partial), group):
            acc = (acc + (a[g:g + group].astype(np.float64).T @ b[g:g + group].astype(np.float64)).astype(np.float32)).astype(np.float32)
        total += acc.astype(np.float64)
    return total


def main():
    rng = np.random.default_rng(7)
    npx, nrow = 307200 // 8, 24
    scale = np.concatenate([np.full(6, 2e2), [1.0, 1.0], np.full(16, 1e-1)]).astype(np.float32)
    z = (rng.standard_normal((npx, nrow)) * scale * np.exp(rng.standard_normal((npx, 1)))).astype(np.float32)
    z[rng.random(npx) < 0.02] = 0.0
    ref = z.astype(np.float64).T @ z.astype(np.float64)
    smax = np.abs(ref).max()
    dscale = np.sqrt(np.outer(np.diag(ref), np.diag(ref)))   # per-entry scale: sqrt(Z_ii Z_jj)

    def report(name, Z):
        e = np.abs(Z.astype(np.float64) - ref)
        print(f"{name:8s} max |err| / max |Z| = {e.max() / smax:.2e}    max |err_ij| / sqrt(Z_ii Z_jj) = {(e / dscale).max():.2e}")

    # chain: sequential fp32 fma over pixels (numpy: accumulate in fp32 in blocks of 4 pixels, exact inside a block)
    report("chain", gram_groups(z, z, group=4))
    h = bf16_round(z); r = (z - h).astype(np.float32); m = bf16_round(r); l = bf16_round((r - m).astype(np.float32))
    assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), z)
    Z3 = gram_groups(h, h) + gram_groups(h, m) + gram_groups(m, h) + gram_groups(h, l) + gram_groups(l, h) + gram_groups(m, m)
    report("bf16x3", Z3)
    report("bf16x2", gram_groups(h, h) + gram_groups(h, m) + gram_groups(m, h))
    # fp16 pieces: rows scaled so that the largest entry of a row is ~2^10 (as a per-pair power of two would)
    k = np.floor(np.log2(1024.0 / np.abs(z).max(axis=0)))
    zs = (z * np.exp2(k)).astype(np.float32)
    hf = zs.astype(np.float16); lf = (zs - hf.astype(np.float32)).astype(np.float16)
    un = np.exp2(-k)[:, None] * np.exp2(-k)[None, :]
    Zf = (gram_groups(hf.astype(np.float32), hf.astype(np.float32)) + gram_groups(hf.astype(np.float32), lf.astype(np.float32)) +
          gram_groups(lf.astype(np.float32), hf.astype(np.float32))) * un
    report("f16x2", Zf)
    ovf = int(np.isinf(hf.astype(np.float32)).sum())
    print(f"f16x2: {ovf} overflows with the per-row scale; without it (raw values): {int(np.isinf(z.astype(np.float16).astype(np.float32)).sum())} of {z.size} values exceed fp16's range")


if __name__ == "__main__":
    main()

"""Multi-GPU layer of the alignment path: one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" in the CPU tests).

Keyframe->frame pairs are independent units (PhotometricFactor::RunAlignmentStep takes only that pair's buffers,
photometric_factor.cpp:267-274), so the pair list is sharded contiguously over ranks with no data-path collective.
The single exchange step is the reduction of the Gauss-Newton normal-equation blocks: every rank scatter-adds its
pairs' 44x44 systems into the block-tridiagonal system of the frame chain exactly as PhotometricFactor::linearize
slices JtJ/Jtr into G11..G33 / g1..g3 (photometric_factor.cpp:105-161), then one all-reduce sums the ranks' partial
systems (fixed pair->rank map, so the result is independent of the world size up to fp32 summation of at most two
contributions per block).  The reference itself has no multi-GPU path (single GPU, default stream)."""
import numpy as np
import torch


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of `n_items` for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class NormalEquations:
    """Block-tridiagonal normal equations over a chain of F frames; frame k carries (pose 6, code CS).  Pair k links
    keyframe k -> frame k+1 and contributes at (pose_k, pose_{k+1}, code_k).  Storage: H[F][2][D][D] (diagonal block,
    upper off-diagonal block to frame k+1) and g[F][D], D = 6 + CS.  assemble() is a gather + index_add on `device`."""

    def __init__(self, n_frames, cs, device):
        D = 6 + cs
        NP = 12 + cs
        self.D, self.F, self.NP, self.cs = D, n_frames, NP, cs
        # one flat exchange buffer (H then g): the multi-GPU reduction is a single collective
        nH = n_frames * 2 * D * D
        self.buf = torch.zeros(nH + n_frames * D, dtype=torch.float32, device=device)
        self.H = self.buf[:nH].view(n_frames, 2, D, D)
        self.g = self.buf[nH:].view(n_frames, D)
        nt = NP * (NP + 1) // 2
        iu = np.triu_indices(NP)
        packed = np.zeros((NP, NP), np.int64)
        packed[iu] = np.arange(nt)
        packed = packed + np.triu(packed, 1).T          # full symmetric -> packed upper-triangular index
        # item parameter n -> (frame offset, local index): pose0 -> (0, 0..5), pose1 -> (1, 0..5), code0 -> (0, 6..)
        fo = np.array([0] * 6 + [1] * 6 + [0] * cs)
        li = np.array(list(range(6)) + list(range(6)) + list(range(6, 6 + cs)))
        src, dst = [], []
        Hs = 2 * D * D
        for a in range(NP):
            for b in range(NP):
                fa, fb = fo[a], fo[b]
                if fa == fb:
                    off = fa * Hs + li[a] * D + li[b]
                elif fa == 0 and fb == 1:
                    off = D * D + li[a] * D + li[b]
                else:
                    continue   # the lower off-diagonal block is the transpose of the stored one
                src.append(packed[a, b]); dst.append(off)
        self.src = torch.tensor(src, dtype=torch.int64, device=device)
        self.dst = torch.tensor(dst, dtype=torch.int64, device=device)
        self.gsrc = torch.tensor(nt + np.arange(NP), dtype=torch.int64, device=device)
        self.gdst = torch.tensor(fo * D + li, dtype=torch.int64, device=device)
        self.Hs = Hs

    def assemble(self, items_u8, first_frame, n_pairs, item_size):
        """items_u8: uint8 tensor with `n_pairs` JTJJrReductionItem<float,12+CS> records (device memory of this rank)."""
        f = items_u8.view(torch.float32).view(n_pairs, item_size // 4)
        base = torch.arange(n_pairs, device=f.device, dtype=torch.int64) + first_frame
        self.H.zero_()
        self.g.zero_()
        self.H.view(-1).index_add_(0, (base[:, None] * self.Hs + self.dst[None, :]).reshape(-1), f[:, self.src].reshape(-1))
        self.g.view(-1).index_add_(0, (base[:, None] * self.D + self.gdst[None, :]).reshape(-1), f[:, self.gsrc].reshape(-1))

    def assemble_native(self, ctx, items_u8, first_frame, n_pairs):
        """Same as assemble(), as ONE libdfx kernel on the context's stream (dfx_neq_assemble_async); GPU only."""
        import ctypes as C
        from . import _lib
        _lib.check(_lib.lib().dfx_neq_assemble_async(ctx.handle, self.cs, C.c_void_p(items_u8.data_ptr()), int(n_pairs), int(first_frame),
                                                     self.F, C.c_void_p(self.H.data_ptr()), C.c_void_p(self.g.data_ptr()), 1))

    def all_reduce(self, dist):
        """Exchange step, replicated result: sum the ranks' partial systems on every rank (RCCL all-reduce over xGMI)."""
        dist.all_reduce(self.buf)

    def reduce(self, dist, root=0):
        """Exchange step, as the solver needs it: sum the ranks' partial systems onto `root`, where the (sequential) solve runs --
        half the xGMI traffic of an all-reduce.  The buffers of the other ranks are left as they were (partial)."""
        dist.reduce(self.buf, dst=root)

    def dense(self):
        """Full symmetric (F*D) x (F*D) matrix -- for tests / small systems only."""
        n = self.F * self.D
        M = torch.zeros((n, n), dtype=torch.float64)
        H = self.H.detach().cpu().double()
        for k in range(self.F):
            s = slice(k * self.D, (k + 1) * self.D)
            M[s, s] += H[k, 0]
            if k + 1 < self.F:
                s2 = slice((k + 1) * self.D, (k + 2) * self.D)
                M[s, s2] += H[k, 1]
                M[s2, s] += H[k, 1].T
        return M

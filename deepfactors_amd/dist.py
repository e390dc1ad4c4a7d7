"""Multi-GPU layer of the alignment path: one process per GPU (torch.distributed; backend
"nccl" is RCCL on ROCm, "gloo" in the CPU tests).

Keyframe->frame pairs are independent units (PhotometricFactor::RunAlignmentStep takes only that pair's buffers,
photometric_factor.cpp:267-274), so the pair list is sharded contiguously over ranks with no data-path collective.
The single exchange step carries the pairs' 44x44 systems to the rank that solves, in one of two forms:

  reduce mode   every rank sums ITS pairs into the block-sparse normal equations of the keyframe graph exactly as
                PhotometricFactor::linearize slices JtJ/Jtr into G11..G33 / g1..g3 (photometric_factor.cpp:105-161), then ONE
                RCCL reduce (or all-reduce) adds the ranks' flat buffers;
  gather mode   the ranks all-gather their items (4152 B per pair at CS = 32: exactly what the reference hands to one
                gtsam::HessianFactor per pair, photometric_factor.cpp:180) and the system is assembled from all of them in
                ascending pair order -- bit-identical for every world size; it is also what a host needs to emit the per-pair
                factors unchanged.

The reference itself has no multi-GPU path (single GPU, default stream) and links arbitrary keyframe -> frame pairs in both
directions (mapper.cpp:308-311); PairGraph is that structure."""
import ctypes as C

import numpy as np
import torch


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of `n_items` for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class PairGraph:
    """n_nodes keyframes / frames, pairs[p] = (keyframe node, frame node).  Node n owns (pose 6, code CS)."""

    def __init__(self, n_nodes, pairs):
        self.n_nodes = int(n_nodes)
        self.pairs = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
        if len(self.pairs) == 0 or self.pairs.min() < 0 or self.pairs.max() >= self.n_nodes or np.any(self.pairs[:, 0] == self.pairs[:, 1]):
            raise ValueError("pairs must link two distinct nodes in [0, n_nodes)")

    @property
    def n_pairs(self):
        return len(self.pairs)

    @staticmethod
    def chain(n_pairs):
        """Pair p links keyframe p -> frame p + 1 (a trajectory; block-tridiagonal system)."""
        return PairGraph(n_pairs + 1, [(p, p + 1) for p in range(n_pairs)])

    @staticmethod
    def window(n_keyframes, neighbours):
        """Every keyframe linked (as the pair's keyframe) to its `neighbours` nearest keyframes by index, on both sides: BASELINE
        configs[3] is window(64, 16) = 1024 directed pairs.  Pairs are ordered by source keyframe, so a contiguous shard of the
        pair list re-uses its keyframes' Jacobians."""
        pairs = []
        for i in range(n_keyframes):
            cand = sorted((j for j in range(n_keyframes) if j != i), key=lambda j: (abs(j - i), j))[:neighbours]
            pairs.extend((i, j) for j in sorted(cand))
        return PairGraph(n_keyframes, pairs)

    @staticmethod
    def all_pairs(n_keyframes, both_directions=False):
        """Every pair i < j (BASELINE configs[2]: 16 keyframes -> 120 pairs), optionally also j -> i as the mapper links them
        (mapper.cpp:308-311)."""
        pairs = [(i, j) for i in range(n_keyframes) for j in range(n_keyframes) if (i < j or (both_directions and i != j))]
        return PairGraph(n_keyframes, pairs)


class NormalEquations:
    """Block-sparse Gauss-Newton normal equations over a PairGraph in ONE flat float32 buffer (the unit of the RCCL exchange):
    Hd [n_nodes][D][D], Ho [n_pairs][D][6], g [n_nodes][D], D = 6 + CS (layout of include/dfx.h, dfx_graph_*).
    assemble() is the torch formulation (any device; the CPU tests and the reference for the native kernel),
    assemble_native() the libdfx kernel."""

    def __init__(self, graph, cs, device):
        self.graph, self.cs = graph, int(cs)
        D, NP = 6 + self.cs, 12 + self.cs
        self.D, self.NP = D, NP
        K, P = graph.n_nodes, graph.n_pairs
        nd, no = K * D * D, P * D * 6
        self.buf = torch.zeros(nd + no + K * D, dtype=torch.float32, device=device)
        self.Hd = self.buf[:nd].view(K, D, D)
        self.Ho = self.buf[nd:nd + no].view(P, D, 6)
        self.g = self.buf[nd + no:].view(K, D)
        self._native = None
        nt = NP * (NP + 1) // 2
        self.nt = nt
        iu = np.triu_indices(NP)
        packed = np.zeros((NP, NP), np.int64)
        packed[iu] = np.arange(nt)
        packed = packed + np.triu(packed, 1).T                       # full symmetric -> packed upper-triangular index
        kf = np.array(list(range(6)) + list(range(12, 12 + self.cs)))   # node-local index -> item parameter, keyframe role
        fr = np.arange(6, 12)                                        # frame role: pose1
        dev = torch.device(device)
        self._kk = torch.tensor(packed[np.ix_(kf, kf)].reshape(-1), dtype=torch.int64, device=dev)    # D*D
        self._ff = torch.tensor(packed[np.ix_(fr, fr)].reshape(-1), dtype=torch.int64, device=dev)    # 36
        self._kf = torch.tensor(packed[np.ix_(kf, fr)].reshape(-1), dtype=torch.int64, device=dev)    # D*6
        self._gk = torch.tensor(nt + kf, dtype=torch.int64, device=dev)
        self._gf = torch.tensor(nt + fr, dtype=torch.int64, device=dev)
        self._pairs = torch.tensor(graph.pairs.astype(np.int64), device=dev)
        ff_dst = (np.arange(6)[:, None] * D + np.arange(6)[None, :]).reshape(-1)
        self._ff_dst = torch.tensor(ff_dst, dtype=torch.int64, device=dev)

    # ---- assembly -------------------------------------------------------------------------------------------------------
    def assemble(self, items_u8, first_pair, n_local, item_size):
        """items_u8: uint8 tensor with the JTJJrReductionItem<float,12+CS> records of the pairs [first_pair, first_pair + n_local).
        Overwrites the whole buffer with the contribution of these pairs (accumulated in double, ascending pair order)."""
        D = self.D
        f = items_u8.view(torch.float32).view(n_local, item_size // 4).double()
        pr = self._pairs[first_pair:first_pair + n_local]
        Hd = torch.zeros(self.graph.n_nodes, D * D, dtype=torch.float64, device=self.buf.device)
        g = torch.zeros(self.graph.n_nodes, D, dtype=torch.float64, device=self.buf.device)
        Hd.index_add_(0, pr[:, 0], f[:, self._kk])
        Hd.view(-1).index_add_(0, (pr[:, 1, None] * (D * D) + self._ff_dst[None, :]).reshape(-1), f[:, self._ff].reshape(-1))
        g.index_add_(0, pr[:, 0], f[:, self._gk])
        g.view(-1).index_add_(0, (pr[:, 1, None] * D + torch.arange(6, device=g.device)[None, :]).reshape(-1), f[:, self._gf].reshape(-1))
        self.Hd.copy_(Hd.view_as(self.Hd).float())
        self.g.copy_(g.float())
        self.Ho.zero_()
        self.Ho[first_pair:first_pair + n_local] = f[:, self._kf].view(n_local, D, 6).float()

    def native_handle(self, ctx):
        if self._native is None:
            from . import _lib
            h = C.c_void_p()
            _lib.check(_lib.lib().dfx_graph_create(ctx.handle, self.cs, self.graph.n_nodes, self.graph.n_pairs,
                                                   self.graph.pairs.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(h)))
            assert int(_lib.lib().dfx_graph_system_floats(h)) == self.buf.numel()
            self._native = h
        return self._native

    def assemble_native(self, ctx, items_u8, first_pair, n_local):
        """Same result as assemble(), as ONE libdfx kernel on the context's stream (dfx_graph_assemble_async); GPU only."""
        from . import _lib
        _lib.check(_lib.lib().dfx_graph_assemble_async(ctx.handle, self.native_handle(ctx), C.c_void_p(items_u8.data_ptr()), int(first_pair),
                                                       int(n_local), C.c_void_p(self.buf.data_ptr())))

    def __del__(self):
        try:
            if self._native is not None:
                from . import _lib
                _lib.lib().dfx_graph_destroy(self._native)
                self._native = None
        except Exception:
            pass

    # ---- exchange step --------------------------------------------------------------------------------------------------
    def all_reduce(self, dist):
        """Reduce mode, replicated result: sum the ranks' partial systems on every rank (RCCL all-reduce over xGMI)."""
        dist.all_reduce(self.buf)

    def reduce(self, dist, root=0):
        """Reduce mode as the solver needs it: sum the ranks' partial systems onto `root`, where the (sequential) solve runs --
        half the xGMI traffic of an all-reduce.  The buffers of the other ranks are left as they were (partial)."""
        dist.reduce(self.buf, dst=root)

    def reduce_async(self, dist, root=0):
        """reduce() as a non-blocking collective: returns the work handle; `wait()` on it orders the caller's stream (RCCL) or the host
        (gloo) behind the sum.  The buffer must not be written again before that wait."""
        return dist.reduce(self.buf, dst=root, async_op=True)

    @staticmethod
    def gather_items(dist, items_u8, n_total, item_size, world):
        """Gather mode: every rank contributes the items of its contiguous shard (shard_range) and receives all n_total items in
        pair order.  Shards of unequal size are padded to the largest for the collective."""
        per = -(-n_total // world)
        pad = torch.zeros(per * item_size, dtype=torch.uint8, device=items_u8.device)
        pad[: items_u8.numel()] = items_u8.view(-1)
        out = torch.empty(world * per * item_size, dtype=torch.uint8, device=items_u8.device)
        dist.all_gather_into_tensor(out, pad)
        if n_total % world == 0:
            return out
        chunks = []
        for r in range(world):
            lo, hi = shard_range(n_total, r, world)
            chunks.append(out[r * per * item_size: r * per * item_size + (hi - lo) * item_size])
        return torch.cat(chunks)

    # ---- for tests / small systems ----------------------------------------------------------------------------------------
    def dense_from(self, buf_host):
        """dense() and the gradient from a HOST copy of the flat buffer (`buf.cpu()`): (H [K D, K D] float64, g [K D] float64), numpy."""
        D, K, P = self.D, self.graph.n_nodes, self.graph.n_pairs
        b = np.asarray(buf_host, np.float32)
        nd, no = K * D * D, P * D * 6
        Hd = b[:nd].reshape(K, D, D).astype(np.float64)
        Ho = b[nd:nd + no].reshape(P, D, 6).astype(np.float64)
        M = np.zeros((K * D, K * D), np.float64)
        for k in range(K):
            M[k * D:(k + 1) * D, k * D:(k + 1) * D] = Hd[k]
        pr = np.asarray(self.graph.pairs, np.int64).reshape(P, 2)
        if len({(int(a), int(c)) for a, c in pr}) == P:
            # every directed pair once: the [D x 6] blocks of one statement do not overlap each other, so two indexed += place them all
            rows = (pr[:, 0, None] * D + np.arange(D))[:, :, None]          # [P][D][1]
            cols = (pr[:, 1, None] * D + np.arange(6))[:, None, :]          # [P][1][6]
            M[rows, cols] += Ho
            M[cols.transpose(0, 2, 1), rows.transpose(0, 2, 1)] += Ho.transpose(0, 2, 1)
        else:
            for p, (a, c) in enumerate(pr):
                a, c = int(a), int(c)
                M[a * D:(a + 1) * D, c * D:c * D + 6] += Ho[p]
                M[c * D:c * D + 6, a * D:(a + 1) * D] += Ho[p].T
        return M, b[nd + no:].astype(np.float64)

    def dense(self):
        """Full symmetric (n_nodes * D)^2 matrix and the gradient vector."""
        D, K = self.D, self.graph.n_nodes
        M = torch.zeros((K * D, K * D), dtype=torch.float64)
        Hd, Ho = self.Hd.detach().cpu().double(), self.Ho.detach().cpu().double()
        for k in range(K):
            s = slice(k * D, (k + 1) * D)
            M[s, s] += Hd[k]
        for p, (a, b) in enumerate(self.graph.pairs):
            ra, cb = slice(a * D, (a + 1) * D), slice(b * D, b * D + 6)
            M[ra, cb] += Ho[p]
            M[cb, ra] += Ho[p].T
        return M



class Comm:
    """The multi-GPU exchange behind the C ABI (include/dfx.h, dfx_comm_*; deepfactors_amd/csrc/dfx_comm.cpp: RCCL over xGMI, resolved at run time) --
    what a C++ mapper calls.  A communicator is created collectively: rank 0 draws the 128-byte unique id, the host program hands it to the other
    ranks by its own means (here: one broadcast over the torch.distributed process group that launched the ranks), every rank calls dfx_comm_create
    with the context of its GPU.  The collectives only ENQUEUE, on the stream the context's results are complete on (its tail stream in deferred-tail
    mode, else its stream): behind the assembly that fills the buffer, and -- with a tail stream -- beside the next launch's step kernel."""

    def __init__(self, handle, rank, world):
        self._h, self.rank, self.world = handle, int(rank), int(world)

    @staticmethod
    def create(ctx, dist, rank, world, device):
        """Collective over the ranks of `dist` (None: a world of one)."""
        from . import _lib
        L = _lib.lib()
        # the id travels with a status byte: a rank 0 that cannot draw an id (no RCCL to load) tells the others instead of leaving them in the broadcast
        uid = torch.zeros(_lib.DFX_COMM_ID_BYTES + 1, dtype=torch.uint8, device=device)
        err = None
        if rank == 0:
            raw = (C.c_ubyte * _lib.DFX_COMM_ID_BYTES)()
            try:
                _lib.check(L.dfx_comm_get_unique_id(raw))
                uid.copy_(torch.tensor(list(raw) + [1], dtype=torch.uint8))
            except Exception as e:   # noqa: BLE001
                err = e
        if dist is not None and world > 1:
            dist.broadcast(uid, 0)
        host = [int(v) for v in uid.cpu().tolist()]
        if err is not None:
            raise err
        if host[-1] != 1:
            raise _lib.DfxError(_lib.DFX_E_HIP, "rank 0 could not draw a communicator id (dfx_comm_get_unique_id failed there)")
        raw = (C.c_ubyte * _lib.DFX_COMM_ID_BYTES)(*host[:-1])
        h = C.c_void_p()
        _lib.check(L.dfx_comm_create(getattr(ctx, "handle", None), raw, int(rank), int(world), C.byref(h)))
        return Comm(h, rank, world)

    def reduce(self, ctx, buf, root=0):
        """Sum `buf` (float32 tensor: a NormalEquations buffer) over the ranks in place onto `root` (< 0: onto every rank): dfx_comm_reduce_f32_async,
        i.e. what dfx_graph_reduce_async does with the graph's system.  Enqueue only."""
        from . import _lib
        _lib.check(_lib.lib().dfx_comm_reduce_f32_async(getattr(ctx, "handle", None), self._h, C.c_void_p(buf.data_ptr()), buf.numel(), int(root)))

    def all_gather_items(self, ctx, items_local_u8, bytes_per_rank, items_all_u8):
        from . import _lib
        _lib.check(_lib.lib().dfx_items_all_gather_async(getattr(ctx, "handle", None), self._h, C.c_void_p(items_local_u8.data_ptr()), int(bytes_per_rank),
                                                         C.c_void_p(items_all_u8.data_ptr())))

    def broadcast(self, ctx, t, root=0):
        from . import _lib
        _lib.check(_lib.lib().dfx_comm_broadcast_async(getattr(ctx, "handle", None), self._h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), int(root)))

    def close(self):
        if self._h:
            from . import _lib
            _lib.lib().dfx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PipelinedReduce:
    """Reduce-mode exchange of a STREAM of independent batches (bench.py, N > 1): the collective of batch k runs on RCCL's stream beside
    the kernels of batch k + 1.  `systems` are NormalEquations of the same graph used round robin; a system is handed out again only
    after the reduce that last read it has been waited for (stream-side on RCCL: the host never blocks).

        sys = pipe.next()        # assemble this batch's items into sys (enqueue the kernels) ...
        pipe.submit()            # ... then start its reduce onto `root`
        pipe.drain()             # before reading results / at the end of a timed region

    A single Gauss-Newton loop cannot use this (its solve needs the reduced system before the next linearisation); a mapper with
    several windows in flight, or a throughput measurement over independent batches, can."""

    def __init__(self, dist, systems, root=0, stream=None):
        """`stream`: the torch.cuda.Stream the systems are WRITTEN on (the context's tail stream in deferred-tail mode, see
        Context.set_tail_stream); None = the current stream.  The collective is issued, and later waited for, with that stream current:
        RCCL starts behind the assembly enqueued there, and the next writer of the buffer (on the same stream) is ordered behind the sum."""
        self.dist, self.systems, self.root, self.stream = dist, list(systems), int(root), stream
        assert len(self.systems) >= 1
        self.pending = [None] * len(self.systems)
        self.count = 0
        self.cur = None

    def _on_stream(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def next(self):
        b = self.count % len(self.systems)
        self.count += 1
        if self.pending[b] is not None:
            with self._on_stream():
                self.pending[b].wait()
            self.pending[b] = None
        self.cur = b
        return self.systems[b]

    def submit(self):
        assert self.cur is not None and self.pending[self.cur] is None
        with self._on_stream():
            self.pending[self.cur] = self.systems[self.cur].reduce_async(self.dist, self.root)

    def drain(self):
        for b, w in enumerate(self.pending):
            if w is not None:
                with self._on_stream():
                    w.wait()
                self.pending[b] = None

    def last(self):
        """The system of the most recent batch (complete on `root` after drain())."""
        return self.systems[self.cur]

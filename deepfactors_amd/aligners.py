"""Python host-side mirror of the reference's operator interface for the alignment hot path.

Same names, argument order and error behaviour as the reference's C++ classes
(``df::SfmAligner<float,CS>`` cu_sfmaligner.h:50-97, ``df::SE3Aligner<float>`` cu_se3aligner.h:40-88,
``df::DepthAligner`` cu_depthaligner.h, free functions of cu_image_proc.h:27-46); every call goes through the C ABI
of libdfx.so.  PyTorch is plumbing only: it owns the device memory (``torch.Tensor`` on ``cuda``) and the stream.

Poses are 7-vectors ``(qx, qy, qz, qw, tx, ty, tz)`` (``Sophus::SE3f``), cameras 6-vectors ``(fx, fy, u0, v0, w, h)``
(``df::PinholeCamera<float>``).  Images are float32 CUDA tensors: ``[H, W]``, gradients ``[H, W, 2]``,
code Jacobians ``[H, W*CS]`` (or ``[H, W, CS]``); rows may be pitched (``stride(0)`` free, inner dims contiguous).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import Cam, CorrItem, DfxError, Img, SE3, SfmPair, SfmParams, check, item_inliers_offset, item_jtj_len, item_size


# ------------------------------------------------------------------------------------------------------------
# result PODs
# ------------------------------------------------------------------------------------------------------------
class JTJJrReductionItem:
    """``df::JTJJrReductionItem<float,NP>`` (reduction_items.h:77-143)."""

    def __init__(self, np_, raw=None):
        self.NP = np_
        if raw is None:
            raw = np.zeros(item_size(np_), np.uint8)
        self.raw = raw
        nt = item_jtj_len(np_)
        f = raw[: (nt + np_ + 1) * 4].view(np.float32)
        self.JtJ = f[:nt]
        self.Jtr = f[nt:nt + np_]
        self.residual = float(f[nt + np_])
        off = item_inliers_offset(np_)
        self.inliers = int(raw[off:off + 8].view(np.uint64)[0])

    def toDenseMatrix(self):
        """``SquareUpperTriangularMatrix::toDenseMatrix()``: mirror the packed upper triangle."""
        M = np.zeros((self.NP, self.NP), np.float32)
        M[np.triu_indices(self.NP)] = self.JtJ
        return M + np.triu(M, 1).T


class CorrespondenceReductionItem:
    """``df::CorrespondenceReductionItem<float>`` (reduction_items.h:35-71)."""

    def __init__(self, residual=0.0, inliers=0):
        self.residual = float(residual)
        self.inliers = int(inliers)


class DenseSfmParams:
    """``df::DenseSfmParams`` (dense_sfm.h:36-43)."""

    def __init__(self, huber_delta=0.1, avg_dpt=2.0, min_dpt=0.0, valid_border=2):
        self.huber_delta, self.avg_dpt, self.min_dpt, self.valid_border = huber_delta, avg_dpt, min_dpt, valid_border

    def _c(self, step_blocks=0):
        return SfmParams(self.huber_delta, self.avg_dpt, self.min_dpt, int(self.valid_border), int(step_blocks))


class SfmAlignerParams:
    """``df::SfmAlignerParams`` (cu_sfmaligner.h:41-48).  Thread counts are accepted for interface parity; the
    gfx950 kernels use fixed 256-thread workgroups and ``step_blocks`` workgroups per pair (0 = automatic)."""

    def __init__(self, sfmparams=None, step_threads=256, step_blocks=0, eval_threads=256, eval_blocks=0):
        self.sfmparams = sfmparams or DenseSfmParams()
        self.step_threads, self.step_blocks = step_threads, step_blocks
        self.eval_threads, self.eval_blocks = eval_threads, eval_blocks


# ------------------------------------------------------------------------------------------------------------
# argument marshalling
# ------------------------------------------------------------------------------------------------------------
class DeviceImage:
    """A device image OWNED THROUGH THE LIBRARY (dfx_img_alloc; include/dfx_host.hpp's ``dfx::DeviceImage``, the device side of the
    reference's ``vc::Image2DManaged`` / SyncedBufferPyramid levels).  Accepted wherever a float32 CUDA tensor is.  What it buys over a
    tensor: the library sees every writer, so a ``valid0`` map kept in one carries a 1-bit-per-pixel shadow and the SfM step stops
    re-reading the map (include/dfx.h, dfx_img_alloc).  `elems_per_px` = 1 (float images; w counts floats) or 2 (gradients)."""

    def __init__(self, ctx, w, h, elems_per_px=1):
        self.ctx, self.elems = ctx, int(elems_per_px)
        self.img = Img()
        check(_lib.lib().dfx_img_alloc(ctx.handle, int(w), int(h), 4 * self.elems, C.byref(self.img)))

    @property
    def shape(self):
        return (int(self.img.h), int(self.img.w)) if self.elems == 1 else (int(self.img.h), int(self.img.w), self.elems)

    def fill(self, value):
        if self.elems != 1:
            raise ValueError("fill is defined for float images")
        check(_lib.lib().dfx_img_fill_f32(self.ctx.handle, C.byref(self.img), float(value)))
        return self

    def upload(self, host):
        a = np.ascontiguousarray(np.asarray(host, np.float32).reshape(self.shape))
        check(_lib.lib().dfx_img_upload(self.ctx.handle, C.byref(self.img), a.ctypes.data_as(C.c_void_p), a.shape[1] * 4 * self.elems, 4 * self.elems))
        return self

    def download(self):
        out = np.empty(self.shape, np.float32)
        check(_lib.lib().dfx_img_download(self.ctx.handle, C.byref(self.img), out.ctypes.data_as(C.c_void_p), out.shape[1] * 4 * self.elems, 4 * self.elems))
        return out

    def valid0_shadow(self):
        """Debug / tests: the map's shadow as a bool array [H, W] ("known to hold 1.0"), or None when it has none."""
        n = (int(self.img.w) * int(self.img.h) + 63) // 64
        words = np.zeros(n, np.uint64)
        got = C.c_size_t(0)
        check(_lib.lib().dfx_debug_read_valid0_shadow(self.ctx.handle, C.byref(self.img), words.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.byref(got)))
        if got.value == 0:
            return None
        bits = np.unpackbits(words.view(np.uint8), bitorder="little")[: int(self.img.w) * int(self.img.h)]
        return bits.reshape(int(self.img.h), int(self.img.w)).astype(bool)

    def free(self):
        if self.img.ptr:
            check(_lib.lib().dfx_img_free(self.ctx.handle, C.byref(self.img)))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _img(t, name, elems_per_px=1):
    if isinstance(t, DeviceImage):
        if t.elems != elems_per_px:
            raise ValueError(f"{name}: image of {t.elems} floats per pixel, expected {elems_per_px}")
        return Img(t.img.ptr, t.img.pitch_bytes, t.img.w, t.img.h)
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise DfxError(_lib.DFX_E_INVALID, f"{name}: tensor must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: float32 required, got {t.dtype}")
    if t.dim() == 3:
        if t.stride(2) != 1 or t.stride(1) != t.shape[2]:
            raise ValueError(f"{name}: inner dimensions must be contiguous")
        h, w = t.shape[0], t.shape[1] * t.shape[2]
    elif t.dim() == 2:
        if t.stride(1) != 1:
            raise ValueError(f"{name}: rows must be contiguous")
        h, w = t.shape
    else:
        raise ValueError(f"{name}: expected a 2-D or 3-D tensor")
    if w % elems_per_px:
        raise ValueError(f"{name}: row length {w} not a multiple of {elems_per_px}")
    return Img(t.data_ptr(), t.stride(0) * 4, w // elems_per_px, h)


def _se3(p):
    a = np.asarray(p, np.float32).reshape(7)
    return SE3((C.c_float * 4)(*a[:4]), (C.c_float * 3)(*a[4:]))


def _cam(c):
    a = np.asarray(c, np.float32).reshape(6)
    return Cam(*[float(v) for v in a])


class Context:
    """One dfx_ctx: device + stream + scratch.  By default it enqueues on the stream that is PyTorch's current stream of `device`
    WHEN THE CONTEXT IS CREATED, so ``torch.cuda.Event`` timing and tensor lifetime rules apply unchanged; inside a later
    ``with torch.cuda.stream(s):`` block call ``use_current_stream()`` (or ``set_stream(s)``) to follow.  `device` None = PyTorch's
    current device.  Every tensor handed to a call must live on the context's device (checked)."""

    def __init__(self, device=None, stream="torch"):
        self._h = C.c_void_p()
        L = _lib.lib()
        if not torch.cuda.is_available():
            raise DfxError(_lib.DFX_E_NOGPU, "no HIP device visible to PyTorch; libdfx has no CPU fallback")
        self.device = int(torch.cuda.current_device() if device is None else device)
        if stream == "torch":
            sh = torch.cuda.current_stream(self.device).cuda_stream
        elif stream is None:
            sh = None
        else:
            sh = int(stream.cuda_stream if isinstance(stream, torch.cuda.Stream) else stream)   # a torch.cuda.Stream or a raw hipStream_t, like set_stream
        check(L.dfx_ctx_create(self.device, C.c_void_p(sh) if sh else None, C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(_lib.lib().dfx_sync(self._h))

    def set_stream(self, stream):
        """Re-bind to another stream of the device (a torch.cuda.Stream, a raw hipStream_t integer, or None = default stream)."""
        sh = stream.cuda_stream if isinstance(stream, torch.cuda.Stream) else stream
        check(_lib.lib().dfx_ctx_set_stream(self._h, C.c_void_p(int(sh)) if sh else None))

    def use_current_stream(self):
        self.set_stream(torch.cuda.current_stream(self.device))

    def set_tail_stream(self, stream):
        """Deferred-tail mode (dfx_set_tail_stream, include/dfx.h): the reduction tail (finalize kernel, graph assembly) of every *_async
        batched step runs on `stream` (a torch.cuda.Stream of this device, or None to switch the mode off), beside the step kernel of the
        next launch.  Results are then complete on THAT stream; `tail_join()` orders the context's stream behind them."""
        self._tail = stream   # keep the torch stream alive
        sh = stream.cuda_stream if isinstance(stream, torch.cuda.Stream) else stream
        check(_lib.lib().dfx_set_tail_stream(self._h, C.c_void_p(int(sh)) if sh else None))

    def tail_join(self):
        check(_lib.lib().dfx_tail_join(self._h))

    def check_device(self, *tensors):
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda and t.device.index != self.device:
                raise DfxError(_lib.DFX_E_INVALID, f"tensor on cuda:{t.device.index} handed to a context of cuda:{self.device}")

    def cu_count(self):
        return int(_lib.lib().dfx_device_cu_count(self._h))

    def alloc_image(self, w, h, elems_per_px=1):
        """A library-owned device image (zero-filled): see DeviceImage."""
        return DeviceImage(self, w, h, elems_per_px)

    def last_mfma_mode(self):
        """_lib.DFX_MFMA_F32_CHAIN / DFX_MFMA_BF16X3: what DFX_MFMA_AUTO (or the pinned mode) resolved to in the last SfM / DepthAligner step."""
        m = C.c_int(0)
        check(_lib.lib().dfx_last_mfma_mode(self._h, C.byref(m)))
        return int(m.value)

    def set_mfma_mode(self, mode):
        """_lib.DFX_MFMA_AUTO (default: the library picks per code size), DFX_MFMA_F32_CHAIN (bitwise an fp32 fmaf chain) or DFX_MFMA_BF16X3
        (exact three-way bf16 split on the bf16 matrix cores, fp32-accurate; see include/dfx.h)."""
        check(_lib.lib().dfx_set_mfma_mode(self._h, int(mode)))

    def set_schedule(self, mode):
        """_lib.DFX_SCHEDULE_AUTO / DFX_SCHEDULE_STATIC (the static, bit-reproducible partition; the default) or
        DFX_SCHEDULE_DYNAMIC (opt-in per-pair item queues for batches of >= 128 pairs); see include/dfx.h."""
        check(_lib.lib().dfx_set_schedule(self._h, int(mode)))

    def set_result_wait(self, mode):
        """_lib.DFX_WAIT_POLL (default: blocking single-result calls poll the word their last kernel stores behind the result) or
        _lib.DFX_WAIT_STREAM (hipStreamSynchronize); see include/dfx.h."""
        check(_lib.lib().dfx_set_result_wait(self._h, int(mode)))

    def configure(self, option, value):
        """dfx_ctx_configure: _lib.DFX_OPT_SIMPLE_DESC_ZEROCOPY / _lib.DFX_OPT_STEP_DESC_ZEROCOPY (0 / 1); result bits never change."""
        check(_lib.lib().dfx_ctx_configure(self._h, int(option), int(value)))

    def last_schedule_dynamic(self):
        """True when the last batched SfM step of this context ran on the dynamic item queues."""
        d = C.c_int(0)
        check(_lib.lib().dfx_last_schedule(self._h, C.byref(d)))
        return bool(d.value)

    def set_profiling(self, enable):
        check(_lib.lib().dfx_set_profiling(self._h, int(bool(enable))))

    def profile_read(self):
        """(n_launches, total_ms) of the step kernel since the last read (HIP events on the context's stream)."""
        n, ms = C.c_int(0), C.c_double(0.0)
        check(_lib.lib().dfx_profile_read(self._h, C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def profile_read_ex(self):
        """(n_launches, total_ms, min_ms, max_ms) of the step kernel since the last read."""
        n, ms, lo, hi = C.c_int(0), C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
        check(_lib.lib().dfx_profile_read_ex(self._h, C.byref(n), C.byref(ms), C.byref(lo), C.byref(hi)))
        return int(n.value), float(ms.value), float(lo.value), float(hi.value)

    def close(self):
        if self._h:
            _lib.lib().dfx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    """The process-wide context of `device` (None = PyTorch's current device; with one process per GPU that is this rank's GPU)."""
    key = int(torch.cuda.current_device() if device is None else device) if torch.cuda.is_available() else 0
    if key not in _default_ctx:
        _default_ctx[key] = Context(key)
    return _default_ctx[key]


def _ctx_for(ctx, *tensors):
    """Explicit context, else the default context of the device the first CUDA tensor lives on; all tensors must agree."""
    if ctx is None:
        dev = next((t.device.index for t in tensors if isinstance(t, torch.Tensor) and t.is_cuda), None)
        ctx = default_context(dev)
    ctx.check_device(*tensors)
    return ctx


# ------------------------------------------------------------------------------------------------------------
# SfmAligner
# ------------------------------------------------------------------------------------------------------------
class SfmAligner:
    """``df::SfmAligner<float,CS>``."""

    def __init__(self, params=None, code_size=32, ctx=None):
        self.params_ = params or SfmAlignerParams()
        self.CS = int(code_size)
        self.ctx = ctx or default_context()
        self.SetStepThreadsBlocks(self.params_.step_threads, self.params_.step_blocks)

    def _p(self):
        """dfx_sfm_params of this aligner: its own step_blocks travel with every call (two aligners on one context do not alias)."""
        return self.params_.sfmparams._c(self.params_.step_blocks)

    # cu_sfmaligner.cpp:187-203: glog CHECK on bad values -> here an exception
    def SetStepThreadsBlocks(self, threads, blocks):
        if threads % 64:
            raise DfxError(_lib.DFX_E_INVALID, "threads must be a multiple of 64 (the CDNA wavefront)")
        if blocks < 0 or blocks > 65535:
            raise DfxError(_lib.DFX_E_INVALID, "blocks out of range [0, 65535]")
        self.params_.step_threads, self.params_.step_blocks = threads, int(blocks)

    def AutoStepBlocks(self, w, h, n_pairs, distinct_jacobians=False):
        """The launch shape the library would pick for `n_pairs` pairs of w x h (dfx_sfm_auto_step_blocks).  A rank of a sharded job pins the
        shape of the WHOLE pair list (`SetStepThreadsBlocks(256, AutoStepBlocks(w, h, n_total))`): its pairs' items are then the bytes the
        single-process run of the whole list produces."""
        out = C.c_int(0)
        check(_lib.lib().dfx_sfm_auto_step_blocks(self.ctx.handle, self.CS, int(w), int(h), int(n_pairs), int(bool(distinct_jacobians)), C.byref(out)))
        return int(out.value)

    def SetEvalThreadsBlocks(self, threads, blocks):
        if threads % 64:
            raise DfxError(_lib.DFX_E_INVALID, "threads must be a multiple of 64 (the CDNA wavefront)")
        self.params_.eval_threads, self.params_.eval_blocks = threads, blocks

    def RunStep(self, pose0, pose1, code0, cam, img0, img1, dpt0, std0, valid0, prx0_jac, grad1):
        """cu_sfmaligner.cpp:149-185.  `code0` is unused by the kernel (depth is already decoded), as in the reference."""
        del code0
        self.ctx.check_device(img0, img1, dpt0, std0, valid0, prx0_jac, grad1)
        np_ = 12 + self.CS
        raw = np.zeros(item_size(np_), np.uint8)
        p = self._p()
        s0, s1, cm = _se3(pose0), _se3(pose1), _cam(cam)
        i0, i1, d0 = _img(img0, "img0"), _img(img1, "img1"), _img(dpt0, "dpt0")
        jc, g1 = _img(prx0_jac, "prx0_jac"), _img(grad1, "grad1", 2)
        sd = _img(std0, "std0") if std0 is not None else None
        vd = _img(valid0, "valid0") if valid0 is not None else None
        check(_lib.lib().dfx_sfm_step(self.ctx.handle, self.CS, C.byref(s0), C.byref(s1), C.byref(cm), C.byref(p), C.byref(i0),
                                      C.byref(i1), C.byref(d0), C.byref(sd) if sd else None, C.byref(vd) if vd else None,
                                      C.byref(jc), C.byref(g1), raw.ctypes.data_as(C.c_void_p)))
        return JTJJrReductionItem(np_, raw)

    def EvaluateError(self, pose0, pose1, cam, img0, img1, dpt0, std0, grad1):
        """cu_sfmaligner.cpp:120-147."""
        out = CorrItem()
        p = self._p()
        s0, s1, cm = _se3(pose0), _se3(pose1), _cam(cam)
        i0, i1, d0 = _img(img0, "img0"), _img(img1, "img1"), _img(dpt0, "dpt0")
        sd = _img(std0, "std0") if std0 is not None else None
        g1 = _img(grad1, "grad1", 2) if grad1 is not None else None
        check(_lib.lib().dfx_sfm_error(self.ctx.handle, C.byref(s0), C.byref(s1), C.byref(cm), C.byref(p), C.byref(i0), C.byref(i1),
                                       C.byref(d0), C.byref(sd) if sd else None, C.byref(g1) if g1 else None, C.byref(out)))
        return CorrespondenceReductionItem(out.residual, out.inliers)

    def EvaluateErrorBatch(self, pair_array, out_items_dev=None):
        """EvaluateError of n pairs in ONE launch (dfx_sfm_error_batch[_async]): PhotometricFactor::error over a factor set
        (photometric_factor.cpp:61-81,197-216).  `pair_array` from make_pairs (only poses, cam, img0, img1, dpt0 are read).  With
        `out_items_dev` (uint8 CUDA tensor, 16 bytes per pair) it only enqueues; otherwise it returns the items."""
        n = len(pair_array)
        p = self._p()
        if out_items_dev is not None:
            if out_items_dev.numel() * out_items_dev.element_size() < 16 * n:
                raise ValueError("output buffer too small")
            check(_lib.lib().dfx_sfm_error_batch_async(self.ctx.handle, C.byref(p), pair_array, n, C.c_void_p(out_items_dev.data_ptr())))
            return None
        out = (CorrItem * n)()
        check(_lib.lib().dfx_sfm_error_batch(self.ctx.handle, C.byref(p), pair_array, n, out))
        return [CorrespondenceReductionItem(o.residual, o.inliers) for o in out]

    # ---- batched extension (one launch over n independent pairs) ----
    def make_pairs(self, pairs):
        """pairs: iterable of dicts with keys pose0,pose1,cam,img0,img1,dpt0,prx0_jac,grad1[,valid0] -> SfmPair array."""
        pairs = list(pairs)
        arr = (SfmPair * len(pairs))()
        for k, q in enumerate(pairs):
            self.ctx.check_device(q["img0"], q["img1"], q["dpt0"], q["prx0_jac"], q["grad1"], q.get("valid0"))
            arr[k].pose0, arr[k].pose1, arr[k].cam = _se3(q["pose0"]), _se3(q["pose1"]), _cam(q["cam"])
            arr[k].img0, arr[k].img1, arr[k].dpt0 = _img(q["img0"], "img0"), _img(q["img1"], "img1"), _img(q["dpt0"], "dpt0")
            arr[k].prx0_jac, arr[k].grad1 = _img(q["prx0_jac"], "prx0_jac"), _img(q["grad1"], "grad1", 2)
            if q.get("valid0") is not None:
                arr[k].valid0 = _img(q["valid0"], "valid0")
        return arr

    def set_poses(self, arr, k, pose0, pose1):
        arr[k].pose0, arr[k].pose1 = _se3(pose0), _se3(pose1)

    @staticmethod
    def set_poses_all(arr, poses, idx0, idx1):
        """A relinearisation round moves poses, not images: rewrite pose0 = poses[idx0[k]], pose1 = poses[idx1[k]] of EVERY record of a make_pairs()
        array in place (numpy views over the ctypes array, no per-pair Python) -- the image half of the records is marshalled once per graph.
        poses: [K][7] (x y z w tx ty tz), idx0 / idx1: [n] keyframe indices (e.g. the two columns of PairGraph.pairs)."""
        n = len(arr)
        P = np.ascontiguousarray(np.asarray(poses, np.float32).reshape(-1, 7))
        raw = np.frombuffer(arr, dtype=np.uint8).reshape(n, C.sizeof(SfmPair))
        for field, idx in ((SfmPair.pose0, idx0), (SfmPair.pose1, idx1)):
            raw[:, field.offset:field.offset + 28] = P[np.asarray(idx, np.int64)].view(np.uint8).reshape(n, 28)

    def RunStepBatchAsync(self, pair_array, out_items_dev):
        """Enqueue one batched launch; results land in `out_items_dev` (uint8 CUDA tensor, n*item_size bytes)."""
        n = len(pair_array)
        isz = item_size(12 + self.CS)
        if out_items_dev.numel() * out_items_dev.element_size() < n * isz:
            raise ValueError("output buffer too small")
        p = self._p()
        check(_lib.lib().dfx_sfm_step_batch_async(self.ctx.handle, self.CS, C.byref(p), pair_array, n,
                                                  C.c_void_p(out_items_dev.data_ptr())))

    def RunStepBatchAssembleAsync(self, pair_array, out_items_dev, neq, first_pair, fused=True):
        """RunStepBatchAsync plus the assembly of this rank's items into the keyframe graph's block-sparse normal equations
        (deepfactors_amd.dist.NormalEquations over a PairGraph): `pair_array` holds the graph's pairs [first_pair, first_pair + n).
        One call (dfx_sfm_step_batch_assemble_async: the assembly runs inside the launch's reduction-tail kernel); `fused=False` issues
        the two calls dfx_sfm_step_batch_async + dfx_graph_assemble_async instead -- same bits.  Enqueue only."""
        if neq.cs != self.CS:
            raise ValueError("normal-equation buffer has a different code size")
        if not fused:
            self.RunStepBatchAsync(pair_array, out_items_dev)
            neq.assemble_native(self.ctx, out_items_dev, int(first_pair), len(pair_array))
            return
        n = len(pair_array)
        if out_items_dev.numel() * out_items_dev.element_size() < n * item_size(12 + self.CS):
            raise ValueError("output buffer too small")
        p = self._p()
        check(_lib.lib().dfx_sfm_step_batch_assemble_async(self.ctx.handle, self.CS, C.byref(p), pair_array, n, C.c_void_p(out_items_dev.data_ptr()),
                                                          neq.native_handle(self.ctx), int(first_pair), C.c_void_p(neq.buf.data_ptr())))

    def LinearizeBatch(self, pair_array, prx0_orig, codes0, out_items_dev=None):
        """PhotometricFactor::RunAlignmentStep over a batch (photometric_factor.cpp:225-293): UpdateDepthMaps once per distinct keyframe
        depth map of the batch, then ONE batched RunStep (dfx_sfm_linearize_batch[_async]).  `prx0_orig`: one tensor per pair,
        `codes0`: [n][CS] array.  With `out_items_dev` (uint8 CUDA tensor) it only enqueues; otherwise it returns the items."""
        n = len(pair_array)
        codes = np.ascontiguousarray(np.asarray(codes0, np.float32).reshape(n, self.CS))
        imgs = (Img * n)(*[_img(t, "prx0_orig") for t in prx0_orig])
        self.ctx.check_device(*prx0_orig)
        p = self._p()
        cp = codes.ctypes.data_as(C.POINTER(C.c_float))
        if out_items_dev is not None:
            check(_lib.lib().dfx_sfm_linearize_batch_async(self.ctx.handle, self.CS, C.byref(p), pair_array, imgs, cp, n, C.c_void_p(out_items_dev.data_ptr())))
            return None
        np_ = 12 + self.CS
        isz = item_size(np_)
        raw = np.zeros(n * isz, np.uint8)
        check(_lib.lib().dfx_sfm_linearize_batch(self.ctx.handle, self.CS, C.byref(p), pair_array, imgs, cp, n, raw.ctypes.data_as(C.c_void_p)))
        return [JTJJrReductionItem(np_, raw[k * isz:(k + 1) * isz]) for k in range(n)]

    def RunStepBatch(self, pair_array):
        n = len(pair_array)
        np_ = 12 + self.CS
        isz = item_size(np_)
        raw = np.zeros(n * isz, np.uint8)
        p = self._p()
        check(_lib.lib().dfx_sfm_step_batch(self.ctx.handle, self.CS, C.byref(p), pair_array, n, raw.ctypes.data_as(C.c_void_p)))
        return [JTJJrReductionItem(np_, raw[k * isz:(k + 1) * isz]) for k in range(n)]

    @staticmethod
    def items_from_bytes(raw, cs):
        np_ = 12 + cs
        isz = item_size(np_)
        raw = np.ascontiguousarray(raw).view(np.uint8)
        return [JTJJrReductionItem(np_, raw[k * isz:(k + 1) * isz]) for k in range(len(raw) // isz)]


# ------------------------------------------------------------------------------------------------------------
# SE3Aligner
# ------------------------------------------------------------------------------------------------------------
class SE3Aligner:
    """``df::SE3Aligner<float>``."""

    def __init__(self, ctx=None):
        self.ctx = ctx or default_context()
        self.huber_delta_ = 0.1

    def SetHuberDelta(self, val):
        self.huber_delta_ = float(val)

    def RunStep(self, se3, cam, img0, img1, dpt0, grad1):
        """cu_se3aligner.cpp:153-176."""
        raw = np.zeros(item_size(6), np.uint8)
        s, cm = _se3(se3), _cam(cam)
        i0, i1, d0, g1 = _img(img0, "img0"), _img(img1, "img1"), _img(dpt0, "dpt0"), _img(grad1, "grad1", 2)
        check(_lib.lib().dfx_se3_step(self.ctx.handle, C.byref(s), C.byref(cm), C.byref(i0), C.byref(i1), C.byref(d0), C.byref(g1),
                                      self.huber_delta_, raw.ctypes.data_as(C.c_void_p)))
        return JTJJrReductionItem(6, raw)

    def make_pairs(self, pairs):
        """pairs: iterable of dicts with keys se3 (pose_10), cam, img0, img1, dpt0, grad1 -> SE3Pair array for RunStepBatch."""
        pairs = list(pairs)
        arr = (_lib.SE3Pair * len(pairs))()
        for k, q in enumerate(pairs):
            self.ctx.check_device(q["img0"], q["img1"], q["dpt0"], q["grad1"])
            arr[k].pose_10, arr[k].cam = _se3(q["se3"]), _cam(q["cam"])
            arr[k].img0, arr[k].img1, arr[k].dpt0 = _img(q["img0"], "img0"), _img(q["img1"], "img1"), _img(q["dpt0"], "dpt0")
            arr[k].grad1 = _img(q["grad1"], "grad1", 2)
        return arr

    def RunStepBatch(self, pair_array, out_items_dev=None):
        """n independent RunStep of one image size in ONE launch (dfx_se3_step_batch[_async]).  With `out_items_dev` (uint8 CUDA tensor,
        120 bytes per pair) it only enqueues; otherwise it returns the items."""
        n = len(pair_array)
        isz = item_size(6)
        if out_items_dev is not None:
            if out_items_dev.numel() * out_items_dev.element_size() < isz * n:
                raise ValueError("output buffer too small")
            check(_lib.lib().dfx_se3_step_batch_async(self.ctx.handle, pair_array, n, self.huber_delta_, C.c_void_p(out_items_dev.data_ptr())))
            return None
        raw = np.zeros(n * isz, np.uint8)
        check(_lib.lib().dfx_se3_step_batch(self.ctx.handle, pair_array, n, self.huber_delta_, raw.ctypes.data_as(C.c_void_p)))
        return [JTJJrReductionItem(6, raw[k * isz:(k + 1) * isz]) for k in range(n)]

    def Warp(self, se3, cam, img0, img1, dpt0, img2):
        """cu_se3aligner.cpp:125-151: renders img1 into frame 0 (`img2`), returns the signed residual sum + inliers."""
        out = CorrItem()
        s, cm = _se3(se3), _cam(cam)
        i0, i1, d0, i2 = _img(img0, "img0"), _img(img1, "img1"), _img(dpt0, "dpt0"), _img(img2, "img2")
        check(_lib.lib().dfx_se3_warp(self.ctx.handle, C.byref(s), C.byref(cm), C.byref(i0), C.byref(i1), C.byref(d0), C.byref(i2),
                                      C.byref(out)))
        return CorrespondenceReductionItem(out.residual, out.inliers)


class DepthAligner:
    """``df::DepthAligner<float,CS>`` (cu_depthaligner.cpp:78-110)."""

    def __init__(self, code_size=32, ctx=None):
        self.CS = int(code_size)
        self.ctx = ctx or default_context()

    def RunStep(self, code, target_dpt, prx_orig, prx_jac, avg_dpt=2.0):
        jc = _img(prx_jac, "prx_jac")
        tg, po = _img(target_dpt, "target_dpt"), _img(prx_orig, "prx_orig")
        if jc.w // tg.w != self.CS:   # CHECK_EQ(codesize, CS) at cu_depthaligner.cpp:90-91
            raise DfxError(_lib.DFX_E_INVALID, "DepthAligner used with a different code size than it was created for")
        raw = np.zeros(item_size(self.CS), np.uint8)
        cd = np.ascontiguousarray(np.asarray(code, np.float32).reshape(self.CS))
        check(_lib.lib().dfx_depth_aligner_step(self.ctx.handle, self.CS, cd.ctypes.data_as(C.POINTER(C.c_float)), C.byref(tg),
                                                C.byref(po), C.byref(jc), float(avg_dpt), raw.ctypes.data_as(C.c_void_p)))
        return JTJJrReductionItem(self.CS, raw)


# ------------------------------------------------------------------------------------------------------------
# cu_image_proc.h free functions
# ------------------------------------------------------------------------------------------------------------
def UpdateDepth(code, prx_orig, prx_jac, avg_dpt, dpt_out, ctx=None):
    """``df::UpdateDepth`` (cu_image_proc.cpp:266-277).  ctx None = the default context of the tensors' device."""
    ctx = _ctx_for(ctx, prx_orig, prx_jac, dpt_out)
    cd = np.ascontiguousarray(np.asarray(code, np.float32).reshape(-1))
    po, jc, do = _img(prx_orig, "prx_orig"), _img(prx_jac, "prx_jac"), _img(dpt_out, "dpt_out")
    if jc.w != po.w * len(cd):
        raise DfxError(_lib.DFX_E_INVALID, f"prx_jac row length {jc.w} != W*CS = {po.w}*{len(cd)}")
    check(_lib.lib().dfx_update_depth(ctx.handle, len(cd), cd.ctypes.data_as(C.POINTER(C.c_float)), C.byref(po), C.byref(jc),
                                      float(avg_dpt), C.byref(do)))


def UpdateDepthBatch(codes, prx_origs, prx_jacs, avg_dpt, dpt_outs, ctx=None):
    """n ``df::UpdateDepth`` of one image size in one launch (dfx_update_depth_batch_async; Mapper::UpdateMap, mapper.cpp:860-888).
    Enqueues only."""
    ctx = _ctx_for(ctx, *prx_origs, *prx_jacs, *dpt_outs)
    n = len(prx_origs)
    cd = np.ascontiguousarray(np.asarray(codes, np.float32).reshape(n, -1))
    cs = cd.shape[1]
    po = (Img * n)(*[_img(t, "prx_orig") for t in prx_origs])
    jc = (Img * n)(*[_img(t, "prx_jac") for t in prx_jacs])
    do = (Img * n)(*[_img(t, "dpt_out") for t in dpt_outs])
    check(_lib.lib().dfx_update_depth_batch_async(ctx.handle, cs, n, cd.ctypes.data_as(C.POINTER(C.c_float)), po, jc, float(avg_dpt), do))


def SobelGradients(img, grad, ctx=None):
    """``df::SobelGradients`` (cu_image_proc.cpp:94-112)."""
    ctx = _ctx_for(ctx, img, grad)
    i, g = _img(img, "img"), _img(grad, "grad", 2)
    check(_lib.lib().dfx_sobel_gradients(ctx.handle, C.byref(i), C.byref(g)))


def GaussianBlurDown(inp, out, ctx=None):
    """``df::GaussianBlurDown`` (cu_image_proc.cpp:166-186)."""
    ctx = _ctx_for(ctx, inp, out)
    i, o = _img(inp, "in"), _img(out, "out")
    check(_lib.lib().dfx_gaussian_blur_down(ctx.handle, C.byref(i), C.byref(o)))


def make_pyramids(pyr_imgs, pyr_grads):
    """The dfx_pyramid array of n frames (see BuildPyramids) -- built once for buffers that are reused frame after frame (a camera's ring of frames): the
    marshalling of 2 L image views per frame is then not paid per call."""
    n = len(pyr_imgs)
    arr = (_lib.Pyramid * n)()
    for k in range(n):
        L = len(pyr_imgs[k])
        if L > _lib.DFX_MAX_PYR_LEVELS or len(pyr_grads[k]) != L:
            raise ValueError("pyramid depth")
        arr[k].levels = L
        for i in range(L):
            arr[k].img[i] = _img(pyr_imgs[k][i], "img")
            if pyr_grads[k][i] is not None:
                arr[k].grad[i] = _img(pyr_grads[k][i], "grad", 2)
    arr._keep = (pyr_imgs, pyr_grads)   # the views hold raw pointers: keep the tensors alive with the array
    return arr


def BuildPyramids(pyr_imgs, pyr_grads=None, ctx=None, blocking=False):
    """``Frame::FillPyramids`` (core/mapping/frame.h:80-94) for n frames in ONE enqueue (dfx_build_pyramid_batch_async): `pyr_imgs[k]` = frame k's
    image pyramid (level 0 = the input, already on the device; levels 1.. are written), `pyr_grads[k]` its gradient pyramid (written; an entry may be
    None to skip that level's gradient, as UploadLiveFrame does for level 0, deepfactors.cpp:620-625).  One launch per pyramid level over all
    frames; same bits as GaussianBlurDown / SobelGradients level by level.  `pyr_imgs` may also be an array from make_pyramids (then `ctx` is required
    unless the default context is meant)."""
    if pyr_grads is None:
        arr = pyr_imgs
        ctx = ctx or default_context()
    else:
        flat = [t for p in pyr_imgs for t in p] + [t for p in pyr_grads for t in p if t is not None]
        ctx = _ctx_for(ctx, *flat)
        arr = make_pyramids(pyr_imgs, pyr_grads)
    n = len(arr)
    if blocking and n == 1:
        check(_lib.lib().dfx_build_pyramid(ctx.handle, arr))
        return
    check(_lib.lib().dfx_build_pyramid_batch_async(ctx.handle, arr, n))
    if blocking:
        ctx.sync()


def SquaredError(buf1, buf2, ctx=None):
    """``df::SquaredError`` (cu_image_proc.cpp:208-240)."""
    ctx = _ctx_for(ctx, buf1, buf2)
    a, b = _img(buf1, "buf1"), _img(buf2, "buf2")
    out = C.c_float(0)
    check(_lib.lib().dfx_squared_error(ctx.handle, C.byref(a), C.byref(b), C.byref(out)))
    return float(out.value)


# ------------------------------------------------------------------------------------------------------------
# CameraTracker (core/system/camera_tracker.{h,cpp}) -- device-resident coarse-to-fine Gauss-Newton
# ------------------------------------------------------------------------------------------------------------
class TrackerConfig:
    """``df::CameraTracker::TrackerConfig`` (camera_tracker.h): pyramid_levels, iterations_per_level (index = level,
    0 = finest; flags ``tracking_iters=5,5,10`` are given coarse-to-fine in data/flags/common.flags:9), huber_delta."""

    def __init__(self, pyramid_levels=3, iterations_per_level=(10, 5, 5), huber_delta=0.1):
        if len(iterations_per_level) != pyramid_levels:   # LOG(FATAL) at camera_tracker.cpp:31-32
            raise DfxError(_lib.DFX_E_INVALID, "CameraTracker config error: iterations_per_level size not equal pyramid_levels")
        self.pyramid_levels, self.iterations_per_level, self.huber_delta = pyramid_levels, tuple(iterations_per_level), huber_delta


class CameraTracker:
    """``df::CameraTracker``: TrackFrame runs the whole schedule on the device (dfx_track_frame); pose_ck is the pose of the
    keyframe in the current camera's frame, exactly the `se3` handed to SE3Aligner::RunStep (camera_tracker.cpp:53)."""

    def __init__(self, camera_pyr, config=None, ctx=None):
        self.config_ = config or TrackerConfig(len(camera_pyr))
        if len(camera_pyr) != self.config_.pyramid_levels:
            raise DfxError(_lib.DFX_E_INVALID, "camera pyramid depth != pyramid_levels")
        self.camera_pyr_ = [np.asarray(c, np.float32) for c in camera_pyr]
        self.ctx = ctx or default_context()
        self.kf_ = None
        self.Reset()

    def Reset(self):
        self.pose_ck_ = np.array([0, 0, 0, 1, 0, 0, 0], np.float32)
        self.inliers_, self.error_ = 0.0, float("inf")

    def SetKeyframe(self, pyr_img, pyr_dpt):
        """Keyframe image and depth pyramids (kf->pyr_img, kf->pyr_dpt), finest level first."""
        self.kf_ = (list(pyr_img), list(pyr_dpt))
        # the keyframe's half of the level records is marshalled once per keyframe, not once per tracked frame
        n = self.config_.pyramid_levels
        self._lv = (_lib.TrackLevel * n)()
        for l in range(n):
            self._lv[l].cam = _cam(self.camera_pyr_[l])
            self._lv[l].img0, self._lv[l].dpt0 = _img(self.kf_[0][l], "kf img"), _img(self.kf_[1][l], "kf dpt")
            self._lv[l].iterations = int(self.config_.iterations_per_level[l])

    def SetPoseEstimate(self, pose_ck):
        self.pose_ck_ = np.asarray(pose_ck, np.float32).copy()

    def TrackFrame(self, pyr_img1, pyr_grad1):
        if self.kf_ is None:
            raise RuntimeError("Calling CameraTracker::TrackFrame before a keyframe was set")   # camera_tracker.cpp:44-45
        n, lv = self.config_.pyramid_levels, self._lv
        for l in range(n):
            lv[l].img1, lv[l].grad1 = _img(pyr_img1[l], "img1"), _img(pyr_grad1[l], "grad1", 2)
        res = _lib.TrackResult()
        s = _se3(self.pose_ck_)
        check(_lib.lib().dfx_track_frame(self.ctx.handle, C.byref(s), lv, n, float(self.config_.huber_delta), C.byref(res)))
        self.pose_ck_ = np.array(list(res.pose_ck.q) + list(res.pose_ck.t), np.float32)
        self.inliers_, self.error_ = float(res.inliers_frac), float(res.error)
        self.last_result_ = res
        return self.pose_ck_

    def GetInliers(self):
        return self.inliers_

    def GetError(self):
        return self.error_

    def TrackFrameBatch(self, keyframes, pyr_img1, pyr_grad1, poses_ck=None):
        """Track ONE live frame against N keyframes at once (dfx_track_frame_batch).  `keyframes` = [(pyr_img, pyr_dpt), ...];
        every tracker starts from `poses_ck[k]` (identity when None, as after Reset()).  Returns the N TrackResult records.
        This is the body of the loops in DeepFactors::Relocalize (deepfactors.cpp:713-743) and of the loop detector's geometry
        check (loop_detector.cpp:146-167), which the reference runs one keyframe after the other."""
        n, nl = len(keyframes), self.config_.pyramid_levels
        lv = (_lib.TrackLevel * (n * nl))()
        for k, (kimg, kdpt) in enumerate(keyframes):
            for l in range(nl):
                e = lv[k * nl + l]
                e.cam = _cam(self.camera_pyr_[l])
                e.img0, e.dpt0 = _img(kimg[l], "kf img"), _img(kdpt[l], "kf dpt")
                e.img1, e.grad1 = _img(pyr_img1[l], "img1"), _img(pyr_grad1[l], "grad1", 2)
                e.iterations = int(self.config_.iterations_per_level[l])
        poses = (_lib.SE3 * n)()
        for k in range(n):
            poses[k] = _se3(np.array([0, 0, 0, 1, 0, 0, 0], np.float32) if poses_ck is None else poses_ck[k])
        res = (_lib.TrackResult * n)()
        check(_lib.lib().dfx_track_frame_batch(self.ctx.handle, n, poses, lv, nl, float(self.config_.huber_delta), res))
        return list(res)

    def Relocalize(self, keyframes, pyr_img1, pyr_grad1):
        """DeepFactors::Relocalize (deepfactors.cpp:713-743): reset, track against every keyframe, keep the smallest error.
        Returns (best index, pose_ck of that keyframe); the tracker is left configured on it, as the reference leaves it."""
        res = self.TrackFrameBatch(keyframes, pyr_img1, pyr_grad1)
        best = min(range(len(res)), key=lambda k: res[k].error)
        r = res[best]
        self.SetKeyframe(*keyframes[best])
        self.pose_ck_ = np.array(list(r.pose_ck.q) + list(r.pose_ck.t), np.float32)
        self.inliers_, self.error_ = float(r.inliers_frac), float(r.error)
        self.last_result_ = r
        return best, self.pose_ck_


# ------------------------------------------------------------------------------------------------------------
# SparseGeometricFactor (core/gtsam/sparse_geometric_factor.{h,cpp})
# ------------------------------------------------------------------------------------------------------------
class SparseGeometricFactor:
    """The data-parallel part of ``df::SparseGeometricFactor<float,CS>``: linearize() returns the [N][12 + 2 CS + 1]
    row-major Jacobian rows [A_pose0 | A_pose1 | A_code0 | A_code1 | b] the reference packs into a gtsam::JacobianFactor
    (keys pose0, pose1, code0, code1); error() = 0.5 * sum b^2 (sparse_geometric_factor.cpp:90-142, with the validity
    check of linearize applied)."""

    def __init__(self, cam, points, kf0, kf1, huber_delta, code_size=32, avg_dpt=2.0, ctx=None):
        """kf0 = dict(prx_orig, prx_jac); kf1 = dict(prx_orig, prx_jac, dpt_grad) of device tensors; points = [N][2] ints."""
        self.cam_, self.kf0_, self.kf1_ = np.asarray(cam, np.float32), kf0, kf1
        self.points_ = np.ascontiguousarray(np.asarray(points, np.int32).reshape(-1, 2))
        self.huber_delta_, self.CS, self.avg_dpt_ = float(huber_delta), int(code_size), float(avg_dpt)
        self.ctx = ctx or default_context()

    def linearize(self, pose0, pose1, code0, code1):
        n, nc = len(self.points_), 12 + 2 * self.CS + 1
        rows = np.zeros((n, nc), np.float32)
        c0 = np.ascontiguousarray(np.asarray(code0, np.float32).reshape(self.CS))
        c1 = np.ascontiguousarray(np.asarray(code1, np.float32).reshape(self.CS))
        s0, s1, cm = _se3(pose0), _se3(pose1), _cam(self.cam_)
        p0, j0 = _img(self.kf0_["prx_orig"], "prx0_orig"), _img(self.kf0_["prx_jac"], "prx0_jac")
        p1, j1 = _img(self.kf1_["prx_orig"], "prx1_orig"), _img(self.kf1_["prx_jac"], "prx1_jac")
        dg = _img(self.kf1_["dpt_grad"], "dpt1_grad", 2)
        fp = C.POINTER(C.c_float)
        check(_lib.lib().dfx_sparse_geometric_linearize(self.ctx.handle, self.CS, C.byref(s0), C.byref(s1), c0.ctypes.data_as(fp),
                                                        c1.ctypes.data_as(fp), C.byref(cm), self.points_.ctypes.data_as(C.POINTER(C.c_int32)), n,
                                                        C.byref(p0), C.byref(j0), C.byref(p1), C.byref(j1), C.byref(dg), self.huber_delta_,
                                                        self.avg_dpt_, rows.ctypes.data_as(fp)))
        return rows

    def error(self, pose0, pose1, code0, code1):
        rows = self.linearize(pose0, pose1, code0, code1)
        return 0.5 * float(np.sum(rows[:, -1].astype(np.float64) ** 2))

    def upload_points(self):
        """Keep the factor's points in device memory (the reference samples them once, in the constructor, sparse_geometric_factor.cpp:50-53):
        batched linearisations then copy no point list."""
        self.points_dev_ = torch.from_numpy(self.points_).to(torch.device("cuda", self.ctx.device))
        return self

    def _fill(self, f, pose0, pose1, code0, code1, keep):
        c0 = np.ascontiguousarray(np.asarray(code0, np.float32).reshape(self.CS))
        c1 = np.ascontiguousarray(np.asarray(code1, np.float32).reshape(self.CS))
        keep.extend((c0, c1))
        fp = C.POINTER(C.c_float)
        f.pose0, f.pose1, f.cam = _se3(pose0), _se3(pose1), _cam(self.cam_)
        f.code0, f.code1 = c0.ctypes.data_as(fp), c1.ctypes.data_as(fp)
        dev = getattr(self, "points_dev_", None)
        f.points_xy = C.c_void_p(dev.data_ptr()) if dev is not None else self.points_.ctypes.data_as(C.c_void_p)
        f.n_points, f.points_on_device = len(self.points_), int(dev is not None)
        f.prx0_orig, f.prx0_jac = _img(self.kf0_["prx_orig"], "prx0_orig"), _img(self.kf0_["prx_jac"], "prx0_jac")
        f.prx1_orig, f.prx1_jac = _img(self.kf1_["prx_orig"], "prx1_orig"), _img(self.kf1_["prx_jac"], "prx1_jac")
        f.dpt1_grad = _img(self.kf1_["dpt_grad"], "dpt1_grad", 2)

    @staticmethod
    def prepare(factors):
        """The dfx_sparse_geo_factor array of a factor set with everything that does not change from round to round filled in (camera, decoder images, points):
        `linearize_all(batch, values, ...)` then only writes poses and codes per round."""
        factors = list(factors)
        f0 = factors[0]
        if any(f.CS != f0.CS or f.huber_delta_ != f0.huber_delta_ or f.avg_dpt_ != f0.avg_dpt_ or f.ctx is not f0.ctx for f in factors):
            raise ValueError("the factors of a batch share code size, huber_delta, avg_dpt and context")
        arr = (_lib.SparseGeoFactor * len(factors))()
        zero = np.zeros(f0.CS, np.float32)
        keep = []
        for k, f in enumerate(factors):
            f._fill(arr[k], np.array([0, 0, 0, 1, 0, 0, 0], np.float32), np.array([0, 0, 0, 1, 0, 0, 0], np.float32), zero, zero, keep)
        arr._factors = factors
        arr._codes = np.zeros((len(factors), 2, f0.CS), np.float32)
        fp = C.POINTER(C.c_float)
        for k in range(len(factors)):
            arr[k].code0 = arr._codes[k, 0].ctypes.data_as(fp)
            arr[k].code1 = arr._codes[k, 1].ctypes.data_as(fp)
        return arr

    @staticmethod
    def _marshal(factors, values):
        """(array, factor objects, first factor) for a round: an array from prepare() gets poses and codes only."""
        if hasattr(factors, "_factors"):
            arr, factors = factors, factors._factors
            for k, v in enumerate(values):
                arr[k].pose0, arr[k].pose1 = _se3(v[0]), _se3(v[1])
                arr._codes[k, 0], arr._codes[k, 1] = v[2], v[3]
            return arr, factors, factors[0]
        factors = list(factors)
        f0 = factors[0]
        if any(f.CS != f0.CS or f.huber_delta_ != f0.huber_delta_ or f.avg_dpt_ != f0.avg_dpt_ or f.ctx is not f0.ctx for f in factors):
            raise ValueError("the factors of a batch share code size, huber_delta, avg_dpt and context")
        arr = (_lib.SparseGeoFactor * len(factors))()
        arr._keep = []
        for k, (f, v) in enumerate(zip(factors, values)):
            f._fill(arr[k], *v, arr._keep)
        return arr, factors, f0

    @staticmethod
    def gram_all(factors, values, gram_dev=None):
        """The round's NORMAL EQUATIONS instead of its rows (dfx_sparse_geometric_gram_batch[_async]): per factor the upper triangle (row-major) of
        [A | b]^T [A | b], NC (NC + 1) / 2 floats with NC = 12 + 2 CS + 1 -- what gtsam's elimination forms from the JacobianFactor on the host, formed on the
        device (a 1024-factor round returns 12 MB instead of 157 MB of rows).  `gram_dev`: float32 CUDA tensor [n][NC (NC + 1) / 2] (enqueue only); otherwise
        a host array of that shape is returned.  `gram_dense(G[k], CS)` unpacks one factor's block."""
        arr, factors, f0 = SparseGeometricFactor._marshal(factors, values)
        n, nc = len(factors), 12 + 2 * f0.CS + 1
        ne = nc * (nc + 1) // 2
        if gram_dev is not None:
            if gram_dev.dtype != torch.float32 or gram_dev.numel() < n * ne or not gram_dev.is_contiguous():
                raise ValueError("gram_dev: contiguous float32 CUDA tensor of n x NC (NC + 1) / 2 required")
            check(_lib.lib().dfx_sparse_geometric_gram_batch_async(f0.ctx.handle, f0.CS, arr, n, f0.huber_delta_, f0.avg_dpt_, C.c_void_p(gram_dev.data_ptr())))
            return None
        out = np.zeros((n, ne), np.float32)
        check(_lib.lib().dfx_sparse_geometric_gram_batch(f0.ctx.handle, f0.CS, arr, n, f0.huber_delta_, f0.avg_dpt_, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    @staticmethod
    def gram_dense(packed, code_size):
        """[NC][NC] symmetric matrix from one factor's packed upper triangle (see gram_all); [:-1, :-1] = A^T A, [:-1, -1] = A^T b, [-1, -1] = b^T b."""
        nc = 12 + 2 * int(code_size) + 1
        M = np.zeros((nc, nc), np.float64)
        M[np.triu_indices(nc)] = np.asarray(packed, np.float64)
        return M + np.triu(M, 1).T

    @staticmethod
    def linearize_all(factors, values, rows_dev=None):
        """Every factor of a relinearisation round in ONE launch (dfx_sparse_geometric_linearize_batch[_async]); the reference linearises them one
        after the other inside ISAM2::update.  `factors`: the factor objects, or the array prepare() made of them (the per-round marshalling is then
        poses and codes only).  `values[k]` = (pose0, pose1, code0, code1) of factor k.  With `rows_dev` (float32 CUDA tensor of
        sum(n_points) x (12 + 2 CS + 1)) the rows stay on the device and the call only enqueues; otherwise it returns one host array per factor
        (views of one buffer, one device-to-host copy)."""
        if hasattr(factors, "_factors"):      # an array from prepare(): poses and codes only
            arr, factors = factors, factors._factors
            for k, v in enumerate(values):
                arr[k].pose0, arr[k].pose1 = _se3(v[0]), _se3(v[1])
                arr._codes[k, 0], arr._codes[k, 1] = v[2], v[3]
            n, f0 = len(factors), factors[0]
        else:
            factors = list(factors)
            n = len(factors)
            f0 = factors[0]
            if any(f.CS != f0.CS or f.huber_delta_ != f0.huber_delta_ or f.avg_dpt_ != f0.avg_dpt_ or f.ctx is not f0.ctx for f in factors):
                raise ValueError("the factors of a batch share code size, huber_delta, avg_dpt and context")
            arr = (_lib.SparseGeoFactor * n)()
            keep = []
            for k, (f, v) in enumerate(zip(factors, values)):
                f._fill(arr[k], *v, keep)
        nc = 12 + 2 * f0.CS + 1
        total = sum(len(f.points_) for f in factors)
        if rows_dev is not None:
            if rows_dev.dtype != torch.float32 or rows_dev.numel() < total * nc or not rows_dev.is_contiguous():
                raise ValueError("rows_dev: contiguous float32 CUDA tensor of sum(n_points) x (12 + 2 CS + 1) required")
            check(_lib.lib().dfx_sparse_geometric_linearize_batch_async(f0.ctx.handle, f0.CS, arr, n, f0.huber_delta_, f0.avg_dpt_, C.c_void_p(rows_dev.data_ptr())))
            return None
        rows = np.zeros((total, nc), np.float32)
        check(_lib.lib().dfx_sparse_geometric_linearize_batch(f0.ctx.handle, f0.CS, arr, n, f0.huber_delta_, f0.avg_dpt_, rows.ctypes.data_as(C.POINTER(C.c_float))))
        out, o = [], 0
        for f in factors:
            out.append(rows[o:o + len(f.points_)])
            o += len(f.points_)
        return out

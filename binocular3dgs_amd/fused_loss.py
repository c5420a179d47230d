"""The per-pair loss block of train.py:123-148 as ONE native call (b3gs_binocular_loss):
value and all four pixel gradients in 4 kernel launches instead of ~40 PyTorch ops + their autograd
(MI355X, 800x600, 3 pairs: 10.1 ms -> see tools/loss_time.py).  Same arguments and result as
`loss.binocular_loss` (which stays as the readable PyTorch statement and the parity reference).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


class _Workspace:
    """Per (device, W, H) persistent buffers: the gradients of a pair stay alive until backward uses them, so
    every call gets its own slot from a small ring (the step calls the loss once per pair per iteration)."""
    _cache = {}

    @classmethod
    def get(cls, dev, W, H, slot):
        key = (dev, W, H, slot)
        if key not in cls._cache:
            f = dict(dtype=torch.float32, device=dev)
            n = _lib.lib().b3gs_loss_workspace_floats(W, H)
            # (zeroed once: the library keeps the partial-sum slots at the head of the workspace zero between calls)
            cls._cache[key] = dict(ws=torch.zeros(n, **f), parts=torch.zeros(8, **f), g_image=torch.empty((3, H, W), **f),
                                   g_depth=torch.empty((1, H, W), **f), g_alpha=torch.empty((1, H, W), **f),
                                   g_shifted=torch.empty((3, H, W), **f))
        return cls._cache[key]


class _BinocularLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, alpha, shifted, gt, alpha_weight, focal_x, trans_dist, lambda_dssim, lambda_smooth,
                slot, unit_grad, return_parts, trans_dist_dev=None):
        if not image.is_cuda:
            raise _lib.B3gsError("binocular_loss_fused needs device tensors (no CPU fallback)")
        H, W = image.shape[-2:]
        buf = _Workspace.get(image.device, W, H, slot)
        io = _lib.B3gsLossIO()
        io.W, io.H = W, H
        cont = [t.contiguous() for t in (image, depth, alpha, gt)]
        io.image, io.depth, io.alpha, io.gt_image = (t.data_ptr() for t in cont)
        sh = None if shifted is None else shifted.contiguous()
        aw = None if alpha_weight is None else alpha_weight.contiguous()
        io.shifted_image = None if sh is None else sh.data_ptr()
        io.alpha_weight = None if aw is None else aw.data_ptr()
        io.focal_x, io.trans_dist = float(focal_x or 0.0), float(trans_dist or 0.0)
        io.trans_dist_dev = None if trans_dist_dev is None else trans_dist_dev.data_ptr()
        io.lambda_dssim, io.lambda_smooth, io.grad_scale = float(lambda_dssim), float(lambda_smooth), 1.0
        io.dL_dimage, io.dL_ddepth, io.dL_dalpha = buf["g_image"].data_ptr(), buf["g_depth"].data_ptr(), buf["g_alpha"].data_ptr()
        io.dL_dshifted = buf["g_shifted"].data_ptr()
        io.parts, io.workspace = buf["parts"].data_ptr(), buf["ws"].data_ptr()
        rc = _lib.lib().b3gs_binocular_loss(C.byref(io), torch.cuda.current_stream(image.device).cuda_stream)
        _lib.check(rc, "b3gs_binocular_loss")
        ctx.buf, ctx.has_shift, ctx.unit_grad = buf, sh is not None, bool(unit_grad)
        total = buf["parts"][0].clone()
        if not return_parts:
            return total, None
        parts = buf["parts"].clone()
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        b = ctx.buf
        if ctx.unit_grad:   # the loss is the root of the graph (total.backward()): upstream gradient == 1
            # fresh aliases of the per-slot buffers: when an input is a LEAF (ViewShardedStep forms the loss on detached
            # leaves of the rendered images), AccumulateGrad keeps an incoming tensor nobody else references instead of
            # cloning it -- four 2-6 MB device copies per iteration otherwise.  The buffers are rewritten by the next loss
            # of the same slot: consume the gradients (rasterizer backward) before that, as a step does
            gi, gd, ga, gs = (b[k].view(b[k].shape) for k in ("g_image", "g_depth", "g_alpha", "g_shifted"))
        else:
            gi, gd, ga, gs = (b[k] * g_total for k in ("g_image", "g_depth", "g_alpha", "g_shifted"))
        return (gi, gd, ga, gs if ctx.has_shift else None) + (None,) * 10


def binocular_loss_fused(image, depth, alpha, gt_image, *, lambda_dssim: float = 0.2, shifted_image=None,
                         focal_x: Optional[float] = None, trans_dist: Optional[float] = None, gt_alpha_mask=None,
                         bg_mask=None, lambda_smooth: float = 0.05, slot: int = 0, unit_grad: bool = False,
                         return_parts: bool = False, trans_dist_dev: Optional[torch.Tensor] = None):
    """Drop-in for loss.binocular_loss(...)[0].  `slot`: index of the pair inside the iteration (gradient
    buffers are per slot and reused across iterations).  unit_grad=True skips the multiplication by the
    upstream scalar when the caller does `total.backward()` / sums the pair losses with weight 1.
    return_parts: also return the device tensor [total, Ll1, ssim, l1_masked, smooth, alpha_loss, -, -].
    trans_dist_dev: a 1-element device tensor that replaces `trans_dist` (the shift is drawn anew every iteration,
    train.py:125-126: an iteration replayed as a HIP graph reads it from device memory)."""
    aw = None
    if gt_alpha_mask is not None:
        aw = 1.0 - gt_alpha_mask
    elif bg_mask is not None:
        aw = bg_mask
    total, parts = _BinocularLoss.apply(image, depth, alpha, shifted_image, gt_image, aw, focal_x, trans_dist,
                                        lambda_dssim, lambda_smooth, int(slot), unit_grad, bool(return_parts), trans_dist_dev)
    return (total, parts) if return_parts else total


class _BinocularLossBatch(torch.autograd.Function):
    """All pairs of an iteration in one call: inputs (image, depth, alpha, shifted) x npairs, output = sum of the
    pair losses (+ the per-pair parts [npairs, 8], non-differentiable)."""

    @staticmethod
    def forward(ctx, meta, unit_grad, *tensors):
        n = len(meta)
        L = _lib.lib()
        ios = (_lib.B3gsLossIO * n)()
        bufs, keep, has_shift = [], [], []
        dev = tensors[0].device
        for k, m in enumerate(meta):
            image, depth, alpha, shifted = tensors[4 * k: 4 * k + 4]
            H, W = image.shape[-2:]
            buf = _Workspace.get(dev, W, H, m["slot"])
            cont = [t.contiguous() for t in (image, depth, alpha, m["gt"])]
            sh = None if shifted is None else shifted.contiguous()
            aw = None if m["alpha_weight"] is None else m["alpha_weight"].contiguous()
            keep += cont + [sh, aw]
            io = ios[k]
            io.W, io.H = W, H
            io.image, io.depth, io.alpha, io.gt_image = (t.data_ptr() for t in cont)
            io.shifted_image = None if sh is None else sh.data_ptr()
            io.alpha_weight = None if aw is None else aw.data_ptr()
            io.focal_x, io.trans_dist = float(m["focal_x"] or 0.0), float(m["trans_dist"] or 0.0)
            td = m.get("trans_dist_dev")
            io.trans_dist_dev = None if td is None else td.data_ptr()
            io.lambda_dssim, io.lambda_smooth, io.grad_scale = float(m["lambda_dssim"]), float(m["lambda_smooth"]), 1.0
            io.dL_dimage, io.dL_ddepth, io.dL_dalpha = buf["g_image"].data_ptr(), buf["g_depth"].data_ptr(), buf["g_alpha"].data_ptr()
            io.dL_dshifted = buf["g_shifted"].data_ptr()
            io.parts, io.workspace = buf["parts"].data_ptr(), buf["ws"].data_ptr()
            bufs.append(buf)
            has_shift.append(sh is not None)
        rc = L.b3gs_binocular_loss_batch(n, ios, torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "b3gs_binocular_loss_batch")
        del keep
        ctx.bufs, ctx.has_shift, ctx.unit_grad = bufs, has_shift, bool(unit_grad)
        parts = torch.stack([b["parts"] for b in bufs])
        ctx.mark_non_differentiable(parts)
        return parts[:, 0].sum(), parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        out = []
        for b, hs in zip(ctx.bufs, ctx.has_shift):
            gs = [b["g_image"].view(b["g_image"].shape), b["g_depth"].view(b["g_depth"].shape),
                  b["g_alpha"].view(b["g_alpha"].shape), b["g_shifted"].view(b["g_shifted"].shape) if hs else None]
            if not ctx.unit_grad:
                gs = [None if g is None else g * g_total for g in gs]
            out += gs
        return (None, None) + tuple(out)


def binocular_loss_fused_batch(pairs, lambda_dssim: float = 0.2, lambda_smooth: float = 0.05, unit_grad: bool = False,
                               return_parts: bool = False):
    """Sum of binocular_loss_fused over `pairs` with one launch per stage for all of them.  pairs: list of dicts with
    keys image, depth, alpha, gt_image and optionally shifted_image, focal_x, trans_dist, gt_alpha_mask | bg_mask.
    Pair k uses gradient-buffer slot k."""
    meta, tensors = [], []
    for k, p in enumerate(pairs):
        aw = None
        if p.get("gt_alpha_mask") is not None:
            aw = 1.0 - p["gt_alpha_mask"]
        elif p.get("bg_mask") is not None:
            aw = p["bg_mask"]
        meta.append(dict(gt=p["gt_image"], alpha_weight=aw, focal_x=p.get("focal_x"), trans_dist=p.get("trans_dist"),
                         trans_dist_dev=p.get("trans_dist_dev"),
                         lambda_dssim=p.get("lambda_dssim", lambda_dssim), lambda_smooth=lambda_smooth, slot=k))
        tensors += [p["image"], p["depth"], p["alpha"], p.get("shifted_image")]
    total, parts = _BinocularLossBatch.apply(meta, unit_grad, *tensors)
    return (total, parts) if return_parts else total

"""Drop-in for the reference's `utils/loss_utils.py` (what train.py:16 imports: `l1_loss, ssim, SmoothLoss`):

    l1_loss(network_output, gt, mask=None)                   utils/loss_utils.py:18-21
    l2_loss(network_output, gt)                              :23-24
    ssim(img1, img2, window_size=11, size_average=True)      :36-66
    SmoothLoss().forward(disparity, image)                   :68-91

Same names, arguments and results; each call is ONE launch of libb3gs_raster.so forward and one backward
(`csrc/lossfn.hip`, ABI 9) instead of 3-25 PyTorch kernels, with a gradient for EVERY input that asks for one.  An
unchanged train.py:123-148 that swaps only this import keeps its own PyTorch glue between the calls; the build's own step
uses the one-call block instead (`fused_loss.binocular_loss_fused`).  `loss.py` keeps the PyTorch statement of the same
functions (CPU-capable: the parity reference of the tests).  No CPU fallback here: host tensors raise.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib
from ._cuda import device_guard, raw_stream

_ws = {}      # (device index, stream) -> zeroed float workspace of the scalar reductions (self-cleaning, see the header)


def _workspace(dev: torch.device, stream: int, planes: int, H: int, W: int) -> torch.Tensor:
    need = _lib.lib().b3gs_lossfn_workspace_floats(planes, H, W)
    key = (dev.index, stream)
    w = _ws.get(key)
    if w is None or w.numel() < need:
        w = _ws[key] = torch.zeros(need, dtype=torch.float32, device=dev)
    return w


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.B3gsError(f"{name} is on {t.device}: the loss functions run on the HIP device only (binocular3dgs_amd.loss "
                             "holds the PyTorch statement)")
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


_stream = raw_stream


def _grad_scalar(g: torch.Tensor) -> torch.Tensor:
    return g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()


class _L1(torch.autograd.Function):
    """mean |x*m - y*m| over [batch, channels, hw]; mask None or [batch, hw]."""

    @staticmethod
    def forward(ctx, x, y, mask, batch, channels, hw):
        dev = x.device
        out = torch.empty((), dtype=torch.float32, device=dev)
        s = _stream(dev)
        with device_guard(dev):
            rc = _lib.lib().b3gs_l1_loss_forward(x.data_ptr(), y.data_ptr(), None if mask is None else mask.data_ptr(), batch,
                                                 channels, hw, out.data_ptr(), _workspace(dev, s, 1, 32, 32).data_ptr(), s)
        _lib.check(rc, "b3gs_l1_loss_forward")
        ctx.save_for_backward(x, y, mask)
        ctx.dims = (batch, channels, hw)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y, mask = ctx.saved_tensors
        batch, channels, hw = ctx.dims
        need = ctx.needs_input_grad
        gx = torch.empty_like(x) if need[0] else None
        gy = torch.empty_like(y) if need[1] else None
        gm = torch.empty_like(mask) if (need[2] and mask is not None) else None
        g = _grad_scalar(g)
        with device_guard(x.device):
            rc = _lib.lib().b3gs_l1_loss_backward(x.data_ptr(), y.data_ptr(), None if mask is None else mask.data_ptr(), batch,
                                                  channels, hw, g.data_ptr(), None if gx is None else gx.data_ptr(),
                                                  None if gy is None else gy.data_ptr(), None if gm is None else gm.data_ptr(),
                                                  _stream(x.device))
        _lib.check(rc, "b3gs_l1_loss_backward")
        return gx, gy, gm, None, None, None


def l1_loss(network_output, gt, mask=None):
    """utils/loss_utils.py:18-21: `torch.abs(network_output*mask - gt*mask).mean()` / `torch.abs(network_output - gt).mean()`."""
    x, y = network_output, gt
    if x.shape != y.shape:
        x, y = torch.broadcast_tensors(x, y)
    if mask is not None and mask.shape != x.shape:
        # the reference's use (train.py:135): [1,3,H,W] against a [1,1,H,W] mask -- broadcast over the channel axis in-kernel
        if not (x.dim() >= 3 and mask.dim() == x.dim() and mask.shape[-3] == 1 and mask.shape[:-3] == x.shape[:-3]
                and mask.shape[-2:] == x.shape[-2:]):
            shape = torch.broadcast_shapes(x.shape, mask.shape)
            x, y, mask = x.expand(shape), y.expand(shape), mask.expand(shape)
    x, y = _dev(x, "network_output"), _dev(y, "gt")
    n = x.numel()
    if n == 0:
        return torch.abs(x - y).mean()         # (nan, like the reference)
    if mask is None:
        return _L1.apply(x, y, None, 1, 1, n)
    mask = _dev(mask, "mask")
    if mask.shape == x.shape:
        return _L1.apply(x, y, mask, 1, 1, n)
    hw = x.shape[-1] * x.shape[-2]
    return _L1.apply(x, y, mask, n // (x.shape[-3] * hw), x.shape[-3], hw)


def l2_loss(network_output, gt):
    """utils/loss_utils.py:23-24 (not on the training path)."""
    return ((network_output - gt) ** 2).mean()


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, batch, channels, size_average):
        dev = img1.device
        H, W = img1.shape[-2:]
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        planes = batch * channels
        maps = torch.empty((5 if need2 else 3, planes, H, W), dtype=torch.float32, device=dev) if (need1 or need2) else None
        out = torch.empty(() if size_average else (batch,), dtype=torch.float32, device=dev)
        s = _stream(dev)
        with device_guard(dev):
            rc = _lib.lib().b3gs_ssim_forward(img1.data_ptr(), img2.data_ptr(), batch, channels, H, W, int(size_average),
                                              None if maps is None else maps.data_ptr(), int(need2), out.data_ptr(),
                                              _workspace(dev, s, planes, H, W).data_ptr(), s)
        _lib.check(rc, "b3gs_ssim_forward")
        ctx.save_for_backward(img1, img2, maps)
        ctx.dims = (batch, channels, H, W, bool(size_average))
        return out

    @staticmethod
    def backward(ctx, g):
        img1, img2, maps = ctx.saved_tensors
        batch, channels, H, W, size_average = ctx.dims
        need = ctx.needs_input_grad
        g1 = torch.empty_like(img1) if need[0] else None
        g2 = torch.empty_like(img2) if need[1] else None
        g = _grad_scalar(g)
        with device_guard(img1.device):
            rc = _lib.lib().b3gs_ssim_backward(img1.data_ptr(), img2.data_ptr(), maps.data_ptr(), batch, channels, H, W,
                                               int(size_average), g.data_ptr(), None if g1 is None else g1.data_ptr(),
                                               None if g2 is None else g2.data_ptr(), _stream(img1.device))
        _lib.check(rc, "b3gs_ssim_backward")
        return g1, g2, None, None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:36-66: mean SSIM with an 11x11 Gaussian window (sigma 1.5, zero padding), every channel by
    itself.  img: [C,H,W] or [B,C,H,W]; `size_average=False` -> one value per batch element (4-D input only, as in the
    reference, whose `mean(1).mean(1).mean(1)` has nothing else to reduce)."""
    if window_size != 11:
        raise _lib.B3gsError("ssim: the HIP kernels are built for the reference's window_size=11")
    if img1.shape != img2.shape:
        img1, img2 = torch.broadcast_tensors(img1, img2)
    if img1.dim() not in (3, 4):
        raise ValueError("ssim expects [C,H,W] or [B,C,H,W] images")
    if not size_average and img1.dim() != 4:
        raise IndexError("Dimension out of range (size_average=False needs a [B,C,H,W] input, as in the reference)")
    img1, img2 = _dev(img1, "img1"), _dev(img2, "img2")
    batch = img1.shape[0] if img1.dim() == 4 else 1
    return _Ssim.apply(img1, img2, batch, img1.shape[-3], bool(size_average))


class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disparity, image):
        dev = disparity.device
        B, C, H, W = image.shape
        out = torch.empty((), dtype=torch.float32, device=dev)
        s = _stream(dev)
        with device_guard(dev):
            rc = _lib.lib().b3gs_smooth_loss_forward(disparity.data_ptr(), image.data_ptr(), B, C, H, W, out.data_ptr(),
                                                     _workspace(dev, s, 1, 32, 32).data_ptr(), s)
        _lib.check(rc, "b3gs_smooth_loss_forward")
        ctx.save_for_backward(disparity, image)
        return out

    @staticmethod
    def backward(ctx, g):
        disparity, image = ctx.saved_tensors
        B, C, H, W = image.shape
        need = ctx.needs_input_grad
        gd = torch.empty_like(disparity) if need[0] else None
        gi = torch.empty_like(image) if need[1] else None
        g = _grad_scalar(g)
        with device_guard(image.device):
            rc = _lib.lib().b3gs_smooth_loss_backward(disparity.data_ptr(), image.data_ptr(), B, C, H, W, g.data_ptr(),
                                                      None if gd is None else gd.data_ptr(),
                                                      None if gi is None else gi.data_ptr(), _stream(image.device))
        _lib.check(rc, "b3gs_smooth_loss_backward")
        return gd, gi


class SmoothLoss(nn.Module):
    """utils/loss_utils.py:68-91: edge-aware first-order smoothness of a disparity map.  The reference holds four fixed 3x3
    convolutions (central differences, no padding); here the stencil is in the kernel and the module has no parameters.
    forward(disparity [B,1,H,W], image [B,3,H,W]) -> scalar."""

    def forward(self, disparity, image):
        if disparity.dim() != 4 or image.dim() != 4 or disparity.shape[1] != 1 or disparity.shape[0] != image.shape[0] \
                or disparity.shape[-2:] != image.shape[-2:]:
            raise ValueError(f"SmoothLoss expects disparity [B,1,H,W] and image [B,C,H,W], got {tuple(disparity.shape)} and "
                             f"{tuple(image.shape)}")
        if image.shape[-1] < 3 or image.shape[-2] < 3:
            raise RuntimeError("SmoothLoss: the 3x3 stencil needs at least 3x3 pixels (the reference's convolution raises too)")
        return _Smooth.apply(_dev(disparity, "disparity"), _dev(image, "image"))

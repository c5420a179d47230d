"""Drop-in for the reference's `utils/loss_utils.py` (what train.py:16 imports: `l1_loss, ssim, SmoothLoss`):

    l1_loss(network_output, gt, mask=None)                   utils/loss_utils.py:18-21
    l2_loss(network_output, gt)                              :23-24
    ssim(img1, img2, window_size=11, size_average=True)      :36-66
    SmoothLoss().forward(disparity, image)                   :68-91

Same names, arguments and results; each call is ONE launch of libb3gs_raster.so forward and one backward
(`csrc/lossfn.hip`, ABI 9) instead of 3-25 PyTorch kernels, with a gradient for EVERY input that asks for one.  The
autograd nodes, argument checks and the (locked, per device + stream) reduction workspace are C++ (`csrc/host/loss.cpp`,
the compiled `_C` module); float64 inputs are computed and returned in float32 (the reference would keep float64).  An
unchanged train.py:123-148 that swaps only this import keeps its own PyTorch glue between the calls; the build's own step
uses the one-call block instead (`fused_loss.binocular_loss_fused`).  `loss.py` keeps the PyTorch statement of the same
functions (CPU-capable: the parity reference of the tests).  No CPU fallback here: host tensors raise.
"""
from __future__ import annotations

from torch import nn

from . import _C, _lib  # noqa: F401  (_C: the compiled module, csrc/host/loss.cpp -- the autograd nodes and argument checks)
from .rasterizer import touch_pending      # (a render() output whose forward is still pending is launched by its first use)


def l1_loss(network_output, gt, mask=None):
    """utils/loss_utils.py:18-21: `torch.abs(network_output*mask - gt*mask).mean()` / `torch.abs(network_output - gt).mean()`.
    A [.., 1, H, W] mask against [.., C, H, W] images (the reference's use, train.py:135) is broadcast over the channels
    in-kernel; any other broadcastable mask is expanded."""
    touch_pending(network_output, gt, mask)
    return _C.l1_loss(network_output, gt, mask)


def l2_loss(network_output, gt):
    """utils/loss_utils.py:23-24 (not on the training path)."""
    return ((network_output - gt) ** 2).mean()


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:36-66: mean SSIM with an 11x11 Gaussian window (sigma 1.5, zero padding), every channel by
    itself.  img: [C,H,W] or [B,C,H,W]; `size_average=False` -> one value per batch element (4-D input only, as in the
    reference, whose `mean(1).mean(1).mean(1)` has nothing else to reduce)."""
    touch_pending(img1, img2)
    return _C.ssim(img1, img2, window_size, bool(size_average))


class SmoothLoss(nn.Module):
    """utils/loss_utils.py:68-91: edge-aware first-order smoothness of a disparity map.  The reference holds four fixed 3x3
    convolutions (central differences, no padding); here the stencil is in the kernel and the module has no parameters.
    forward(disparity [B,1,H,W], image [B,3,H,W]) -> scalar."""

    def forward(self, disparity, image):
        touch_pending(disparity, image)
        return _C.smooth_loss(disparity, image)

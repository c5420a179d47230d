"""Drop-in for the pieces of the reference's `utils/graphics_utils.py` the training loop uses
(train.py:19: `from utils.graphics_utils import inverse_warp_images`; scene/cameras.py: the matrix helpers):

    inverse_warp_images(image, disparity, row_indices, column_indices)     utils/graphics_utils.py:80-125
    getWorld2View2 / getProjectionMatrix / fov2focal / focal2fov          :38-77  (camera.py, pinned by golden G3)

`inverse_warp_images` is one launch forward and one backward (`csrc/lossfn.hip`, ABI 9) where the reference loops over
batch and channels in Python (~25 PyTorch kernels per call, index tensors cloned per channel); gradients for the image
(bilinear scatter) and for the disparity.  No CPU fallback: `loss.inverse_warp_images` is the PyTorch statement.
"""
from __future__ import annotations

import torch

from . import _lib
from .camera import focal2fov, fov2focal  # noqa: F401
from .camera import projection_matrix as getProjectionMatrix  # noqa: F401
from ._cuda import device_guard
from .loss_utils import _dev, _stream


def getWorld2View2(R, t, translate=None, scale=1.0):
    """utils/graphics_utils.py:38-49"""
    import numpy as np
    from .camera import world_to_view
    return world_to_view(R, t, np.array([0.0, 0.0, 0.0]) if translate is None else translate, scale)


class _InverseWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, disparity):
        dev = image.device
        B, C, H, W = image.shape
        out = torch.empty_like(image)
        # the gradient of the image is a scatter (atomics): its buffer is zeroed by THIS launch on its way through the
        # pixels, so the backward needs no fill in front of it
        gbuf = torch.empty_like(image) if ctx.needs_input_grad[0] else None
        with device_guard(dev):
            rc = _lib.lib().b3gs_inverse_warp_forward(image.data_ptr(), disparity.data_ptr(), B, C, H, W, out.data_ptr(),
                                                      None if gbuf is None else gbuf.data_ptr(), _stream(dev))
        _lib.check(rc, "b3gs_inverse_warp_forward")
        ctx.save_for_backward(image, disparity)
        ctx.gbuf = gbuf
        return out

    @staticmethod
    def backward(ctx, g):
        image, disparity = ctx.saved_tensors
        B, C, H, W = image.shape
        need = ctx.needs_input_grad
        g = _dev(g, "grad")
        gi = None
        if need[0]:
            gi, ctx.gbuf = ctx.gbuf, None           # (handed to autograd: nothing here keeps a reference to it)
            if gi is None:                          # a second backward through a retained graph
                gi = torch.zeros_like(image)
        gd = torch.empty_like(disparity) if need[1] else None
        with device_guard(image.device):
            rc = _lib.lib().b3gs_inverse_warp_backward(image.data_ptr(), disparity.data_ptr(), g.data_ptr(), B, C, H, W,
                                                       None if gi is None else gi.data_ptr(),
                                                       None if gd is None else gd.data_ptr(), _stream(image.device))
        _lib.check(rc, "b3gs_inverse_warp_backward")
        return gi, gd


def inverse_warp_images(image, disparity, row_indices=None, column_indices=None):
    """utils/graphics_utils.py:80-125: out[b,ch,r,c] = (x1 - d) image[b,ch,r,c+x0] + (d - x0) image[b,ch,r,c+x1] with
    d = disparity[b,0,r,c], x0 = floor(d), x1 = x0 + 1; zero where either column leaves the image.  image [B,C,H,W],
    disparity [B,1,H,W].  `row_indices`, `column_indices` (the reference's meshgrid of pixel coordinates, train.py:61-63)
    are accepted for signature compatibility; the kernel knows where its pixels are."""
    del row_indices, column_indices
    if image.dim() != 4 or disparity.dim() != 4 or disparity.shape[1] != 1 or disparity.shape[0] != image.shape[0] \
            or disparity.shape[-2:] != image.shape[-2:]:
        raise ValueError(f"inverse_warp_images expects image [B,C,H,W] and disparity [B,1,H,W], got {tuple(image.shape)} and "
                         f"{tuple(disparity.shape)}")
    return _InverseWarp.apply(_dev(image, "image"), _dev(disparity, "disparity"))
